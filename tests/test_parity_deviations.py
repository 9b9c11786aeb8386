"""The two DOCUMENTED deviations of the oracle (and therefore of the GPU path) from what a real build of the reference
would do, turned into numbers.  Neither can be pinned here (no OpenCV, the C++ is unbuildable in this
image); both are restated as oracle SWITCHES and counted on the bench's kind of workload:

  (a) matcher ties: the reference returns the first equally-near code its multi-index hash discovers
      (binary_descriptor_matcher.cpp:635-753), the build the lowest index (include/lanefront.h, a-10);
  (b) LSD seed order inside a gradient bin: raster (OpenCV 3.0's per-bin lists; the build) against the order
      libstdc++'s std::sort leaves (the later OpenCV 3.x: std::sort(ordered_points, compare_norm), not stable).

The counts are printed (run with -s) and quoted in DESIGN.md section 2; the assertions only hold what must hold."""
import numpy as np

from lane_slam_amd import default_config, synth
from oracle.oracle import Oracle


def _mih_literal(q, train):
    """Mihasher::query, literally (:635-753): 32 tables of 8-bit substrings, radius s = 0..4, substring k = 0..31, the
    combination loop of :681-741, buckets in insertion order, first index per distance, stop once a code at distance
    s * 32 + k has been seen.  Returns (index, distance) or (-1, -1)."""
    m, b, D, d = 32, 8, 128, 4
    tables = [dict() for _ in range(m)]
    for i, code in enumerate(train):
        for k in range(m):
            tables[k].setdefault(int(code[k]), []).append(i)
    seen, first, numres = set(), {}, [0] * 257
    n = 0
    for s in range(d + 1):
        if n >= 1:
            break
        for k in range(m):
            chunk = int(q[k])
            power = list(range(s)) + [b + 1]
            bit, bitstr = s - 1, 0
            while True:
                if bit != -1:
                    bitstr ^= (1 << power[bit]) if power[bit] == bit else (3 << (power[bit] - 1))
                    power[bit] += 1
                    bit -= 1
                else:
                    for idx in tables[k].get(chunk ^ bitstr, ()):
                        if idx not in seen:
                            seen.add(idx)
                            hd = int(np.unpackbits(train[idx] ^ q).sum())
                            if hd <= D and numres[hd] < 1:
                                first[hd] = idx
                            numres[hd] += 1
                    bit += 1
                    while bit < s and power[bit] == power[bit + 1] - 1:
                        bitstr ^= 1 << (power[bit] - 1)
                        power[bit] = bit
                        bit += 1
                    if bit == s:
                        break
            n += numres[s * m + k]
            if n >= 1:
                break
    for hd in range(D + 1):
        if numres[hd]:
            return first[hd], hd
    return -1, -1


def test_first_discovered_rule_is_restated_correctly():
    o = Oracle(default_config("parity"))
    rng = np.random.default_rng(21)
    train = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (60, 32), dtype=np.uint8)
    # planted ties: several train codes at the SAME distance from a query, differing in different substrings
    for i in range(40):
        dist = int(rng.integers(0, 60))
        for copy in range(int(rng.integers(2, 5))):
            c = q[i].copy()
            for bpos in rng.choice(256, size=dist, replace=False):
                c[bpos >> 3] ^= np.uint8(1 << (bpos & 7))
            train[int(rng.integers(0, 300))] = c
    train[17] = train[250] = q[41]                                # exact duplicates
    idx, dist, ties = o.match_mih(q, train)
    low_idx, low_dist = o.match(q, train)
    assert np.array_equal(dist, low_dist)                         # the distance never depends on the rule
    differ = 0
    for i in range(q.shape[0]):
        wi, wd = _mih_literal(q[i], train)
        assert (idx[i], dist[i]) == (wi, wd), i
        differ += int(idx[i] != low_idx[i])
        if ties[i] == 1:
            assert idx[i] == low_idx[i]
    assert differ > 0 and (ties > 1).sum() >= 30                  # the planted ties do separate the two rules
    assert idx[41] == 17                                          # duplicates share every bucket: train order decides


def test_census_of_the_two_deviations_on_bench_like_input():
    cfg = default_config("fullres")
    o = Oracle(cfg)
    frames = synth.make_batch(24, seed0=0)
    res = [o.process_frame(f, cap=3 * 512) for f in frames]
    codes = np.concatenate([r["code"] for r in res])
    keep = np.concatenate([r["keep"] for r in res])
    # (a) the bench's association: the step's codes against 66 384 random codes (seed 1234) ...
    rand_map = synth.random_codes(66384, 1234)
    q = codes[: 600]
    i_m, d_m, ties = o.match_mih(q, rand_map)
    i_l, d_l = o.match(q, rand_map)
    a_rand = {"queries": int(q.shape[0]), "matched": int((i_l >= 0).sum()), "non_unique_minimum": int((ties > 1).sum()),
              "index_differs": int((i_m != i_l).sum())}
    # ... against a map that also holds the kept segments of OTHER frames of the same scene type (near duplicates) ...
    live = np.concatenate([rand_map[:20000], codes[600:][keep[600:] != 0]])
    i_m, d_m, ties = o.match_mih(q, live)
    i_l, d_l = o.match(q, live)
    assert np.array_equal(d_m, d_l)
    a_live = {"queries": int(q.shape[0]), "matched": int((i_l >= 0).sum()), "non_unique_minimum": int((ties > 1).sum()),
              "index_differs": int((i_m != i_l).sum())}
    # ... and against the bench's own live map, which holds the SAME frames' kept segments from earlier steps (bench.py
    # submits the same batch every step): exact duplicates, the commonest tie of all
    kq = q[keep[:600] != 0]
    again = np.concatenate([rand_map[:20000], kq, kq])
    i_m, d_m, ties = o.match_mih(kq, again)
    i_l, d_l = o.match(kq, again)
    a_same = {"queries": int(kq.shape[0]), "ties_at_distance_0": int(((ties > 1) & (d_l == 0)).sum()), "index_differs": int((i_m != i_l).sum())}
    print("\nmatcher tie census (first-discovered vs lowest index): random map %r; map with other frames' segments %r; map with the "
          "same frames' segments twice %r" % (a_rand, a_live, a_same))
    # exact duplicates share every bucket, so train order = lowest index decides under both rules
    assert a_same["ties_at_distance_0"] == a_same["queries"] and a_same["index_differs"] == 0

    # (b) LSD seed order: frames whose SegmentList changes when seeds inside a bin follow std::sort instead of raster
    changed, total, seg_changed, seg_total = 0, 0, 0, 0
    for geometry, n in (("parity", 30), ("fullres", 6)):
        c2 = default_config(geometry)
        c2["lsd"]["seed_order"] = "opencv30"          # the switch below turns the std::sort order on and off
        o2 = Oracle(c2)
        for f in synth.make_batch(n, seed0=700):
            o2.set_lsd_seed_order(0)
            a = o2.process_frame(f, cap=3 * 512)
            o2.set_lsd_seed_order(1)
            b = o2.process_frame(f, cap=3 * 512)
            o2.set_lsd_seed_order(0)
            same = a["n"] == b["n"] and np.array_equal(a["lines"], b["lines"])
            changed += int(not same)
            total += 1
            seg_total += a["n"]
            if not same:
                sa = {tuple(l) for l in a["lines"].tolist()}
                sb = {tuple(l) for l in b["lines"].tolist()}
                seg_changed += len(sa ^ sb)
    print("LSD seed order census (raster vs libstdc++ std::sort inside a bin): %d of %d frames change, %d of %d segments differ"
          % (changed, total, seg_changed, seg_total))
    assert total == 36
