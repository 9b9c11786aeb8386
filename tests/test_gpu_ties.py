"""The matcher's tie rule on the GPU (VERDICT r3 #1a): among equally near map codes BinaryDescriptorMatcher::match returns the
one Mihasher::query discovers FIRST (ref: src/line_descriptor/src/binary_descriptor_matcher.cpp:635-753).  lf_associate does
that by default (LF_TIE_MIHASHER, k_assoc_ties.hip); the live map on request.  Checked against the oracle's literal
restatement (oracle/lf_oracle_lbd.c: lfo_match_mih, itself checked against a transcription of the reference's loop in
tests/test_parity_deviations.py) -- every index, not only the unique minima."""
from types import SimpleNamespace

import numpy as np
import pytest

from lane_slam_amd import FrontEnd, LineAssociator, default_config, synth

pytestmark = pytest.mark.gpu


def _flip(rng, code, nbits, lo=0, hi=256):
    out = code.copy()
    for b in rng.choice(np.arange(lo, hi), size=nbits, replace=False):
        out[b >> 3] ^= np.uint8(1 << (b & 7))
    return out


def test_planted_ties_follow_the_discovery_order():
    from oracle.oracle import Oracle
    o = Oracle(default_config("parity"))
    fe = FrontEnd(default_config("parity"))
    rng = np.random.default_rng(21)
    nt, nq = 5000, 400
    train = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    # several train codes at the SAME distance from a query, the differing bits in different substrings / of different weights
    for i in range(300):
        d = int(rng.integers(0, 70))
        for _ in range(int(rng.integers(2, 6))):
            train[int(rng.integers(0, nt))] = _flip(rng, q[i], d)
    # ties decided by the substring number alone, by the bit string's place in the enumeration alone, by the index alone
    train[100] = _flip(rng, q[300], 3, 8 * 20, 8 * 21); train[50] = _flip(rng, q[300], 3, 8 * 4, 8 * 5)        # substring 4 before 20
    a = q[301].copy(); a[9] ^= np.uint8(0b00000110); b = q[301].copy(); b[9] ^= np.uint8(0b10000001)
    train[7], train[8] = b, a                                                                                  # same substring, same weight
    train[4000] = train[17] = train[2500] = q[302]                                                             # exact duplicates
    wi, wd, ties = o.match_mih(q, train)
    li, ld = o.match(q, train)
    assert (ties > 1).sum() >= 250 and (wi != li).sum() >= 60          # the planted ties do separate the two rules
    fe.set_tie_rule("mihasher")
    gi, gd = fe.associate(q, train)
    assert np.array_equal(gd, wd)
    assert np.array_equal(gi, wi), "%d of %d indices differ from the reference's rule" % ((gi != wi).sum(), nq)
    assert gi[300] == 50 and gi[302] == 17
    fe.set_tie_rule("lowest")
    gi, gd = fe.associate(q, train)
    assert np.array_equal(gi, li) and np.array_equal(gd, ld)
    fe.close()


def test_the_default_rule_is_the_references():
    from oracle.oracle import Oracle
    o = Oracle(default_config("parity"))
    fe = FrontEnd(default_config("parity"))
    rng = np.random.default_rng(3)
    train = rng.integers(0, 256, (900, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (64, 32), dtype=np.uint8)
    for i in range(64):
        for _ in range(3):
            train[int(rng.integers(0, 900))] = _flip(rng, q[i], 10 + i % 30)
    wi, wd, ties = o.match_mih(q, train)
    li, _ = o.match(q, train)
    assert (wi != li).any()
    gi, gd = fe.associate(q, train)                     # no set_tie_rule call
    assert np.array_equal(gi, wi) and np.array_equal(gd, wd)
    with pytest.raises(ValueError):
        fe.set_tie_rule("newest")
    fe.close()


def test_census_maps_zero_of_600_differ():
    """The three maps of tests/test_parity_deviations.py (the census that counted 71 / 16 / 0 of 600 differing indices for the
    lowest-index rule): under LF_TIE_MIHASHER 0 of 600 differ from the reference's rule on each."""
    from oracle.oracle import Oracle
    cfg = default_config("fullres")
    o = Oracle(cfg)
    fe = FrontEnd(cfg, max_frames=24, max_lines_per_color=512)
    frames = synth.make_batch(24, seed0=0)
    seg = fe.process_batch(frames, describe=True)
    codes, keep = seg.code, seg.keep
    assert codes.shape[0] > 700
    rand_map = synth.random_codes(66384, 1234)
    q = codes[:600]
    live = np.concatenate([rand_map[:20000], codes[600:][keep[600:] != 0]])
    kq = q[keep[:600] != 0]
    again = np.concatenate([rand_map[:20000], kq, kq])
    differ_lowest = []
    for queries, train in ((q, rand_map), (q, live), (kq, again)):
        wi, wd, ties = o.match_mih(queries, train)
        fe.set_tie_rule("mihasher")
        gi, gd = fe.associate(queries, train)
        assert np.array_equal(gd, wd)
        assert np.array_equal(gi, wi), "%d of %d differ" % ((gi != wi).sum(), queries.shape[0])
        fe.set_tie_rule("lowest")
        li, _ = fe.associate(queries, train)
        differ_lowest.append(int((li != wi).sum()))
    print("\nindices that differ from the reference's rule under LF_TIE_LOWEST: %r of 600 / 600 / %d; under LF_TIE_MIHASHER: 0" % (differ_lowest, kq.shape[0]))
    assert differ_lowest[0] > 20 and differ_lowest[2] == 0
    fe.close()


@pytest.mark.parametrize("gating", [False, True])
def test_live_map_with_the_references_rule(gating):
    """lf_map_set_tie_rule(LF_TIE_MIHASHER): association against the live map, sizes across the tile and chunk boundaries,
    colour gating (the discovery order among the entries the query may match), wildcards, and a merge-policy stream where the
    chosen index decides which entry is refreshed."""
    from oracle.oracle import OracleMap
    rng = np.random.default_rng(40 + gating)
    kw = dict(capacity=30000, color_gating=gating, max_distance=128, kept_only=False, tie_rule="mihasher")
    a, o = LineAssociator(**kw), OracleMap(**kw)
    low = OracleMap(**dict(kw, tie_rule="lowest"))
    total, differ = 0, 0
    for k, n in enumerate([1, 63, 64, 2000, 9000, 12000]):
        m = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        mc = rng.integers(0, 3, n).astype(np.uint8)
        if k == 3:
            mc[::5] = 255
        nq = [7, 300, 650][k % 3]
        q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
        qc = rng.integers(0, 3, nq).astype(np.uint8)
        qc[::11] = 255
        for i in range(0, nq, 2):                         # ties, of every colour
            d = int(rng.integers(0, 60))
            for _ in range(int(rng.integers(2, 5))):
                m[int(rng.integers(0, n))] = _flip(rng, q[i], d)
        for x in (a, o, low):
            x.seed(m, mc)
        total += n
        gi, gd = a.associate(q, qc)
        oi, od = o.associate(q, qc)
        li, ld = low.associate(q, qc)
        assert np.array_equal(gd, od) and np.array_equal(od, ld), (k, n)
        assert np.array_equal(gi, oi), (k, n, int((gi != oi).sum()))
        differ += int((oi != li).sum())
    assert differ > 50
    a.close()
    # merge policy: the index decides which entry a segment refreshes
    kw = dict(capacity=3000, color_gating=gating, max_distance=128, policy="merge", merge_distance=40, kept_only=False, tie_rule="mihasher")
    a, o = LineAssociator(**kw), OracleMap(**kw)
    base = rng.integers(0, 256, (40, 32), dtype=np.uint8)
    for step in range(6):
        n = 500
        code = np.stack([_flip(rng, base[int(rng.integers(0, 40))], int(rng.integers(0, 12))) for _ in range(n)])
        color = rng.integers(0, 3, n).astype(np.uint8)
        keep = np.ones(n, np.uint8)
        ground = rng.random((n, 4))
        seg = SimpleNamespace(n=n, frame_offset=np.array([0, n], np.int32), code=code, color=color, keep=keep, ground=ground)
        gi, gd = a.step(seg, None, step)
        oi, od = o.step(code, color, keep, ground, step)
        assert np.array_equal(gd, od) and np.array_equal(gi, oi), step
    sg, so = a.state(), o.state()
    assert sg["size"] == so["size"] and sg["total_refreshed"] == so["total_refreshed"]
    g, r = a.fetch(), o.fetch()
    for k in ("code", "color", "hits", "last_seen"):
        assert np.array_equal(g[k][:so["size"]], r[k][:so["size"]]), k
    a.close()


def test_query_mask_of_the_matcher():
    """match / knnMatch / radiusMatch with a mask (ref: binary_descriptor_matcher.cpp:231-235, 305-309, 477-481): DMatches only for
    the queries whose mask byte is not 0, each with its queryIdx -- lf_select_queries + the unchanged calls."""
    from oracle.oracle import Oracle
    o = Oracle(default_config("parity"))
    fe = FrontEnd(default_config("parity"))
    rng = np.random.default_rng(8)
    m = rng.integers(0, 256, (700, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (2500, 32), dtype=np.uint8)
    mask = (rng.random(2500) < 0.4).astype(np.uint8) * rng.integers(1, 255, 2500).astype(np.uint8)
    sel, qi = fe.select_queries(q, mask)
    keep = np.nonzero(mask)[0]
    assert np.array_equal(qi, keep) and np.array_equal(sel, q[keep])
    gi, gd = fe.associate(sel, m)
    wi, wd, _ = o.match_mih(q, m)
    assert np.array_equal(gi, wi[keep]) and np.array_equal(gd, wd[keep])
    ki, kd = fe.knn_match(sel, m, 3)
    oi, od = o.knn_match(q, m, 3, tie_rule="mihasher")
    assert np.array_equal(ki, oi[keep]) and np.array_equal(kd, od[keep])
    for mk in (np.zeros(2500, np.uint8), np.ones(2500, np.uint8)):
        s2, q2 = fe.select_queries(q, mk)
        assert len(q2) == int(mk.sum()) and np.array_equal(s2, q[mk != 0])
    fe.close()


@pytest.mark.parametrize("nq", [129, 300, 385, 1, 12288, 12289, 13000])
def test_both_shapes_of_the_distance_pass_list_ties_alike(nq):
    """Round 6: the distance pass has a small shape (128 queries per workgroup: associations of up to 12 288 queries) beside the big one
    (256).  Both write the tie pass's lists per 256-query block -- the small shape two pieces per workgroup, its last workgroup also the
    empty pieces an odd number of 128-query blocks leaves.  Ties planted in the LAST queries (the partial block) and the first; every
    index against the oracle's literal Mihasher rule.  (12 289 and 13 000: the big shape.)"""
    from oracle.oracle import Oracle
    o = Oracle(default_config("parity"))
    fe = FrontEnd(default_config("parity"))
    rng = np.random.default_rng(1000 + nq)
    nt = 3000 if nq > 2000 else 20000
    train = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    planted = sorted(set([0, nq - 1, max(nq - 2, 0), nq // 2] + [int(v) for v in rng.integers(0, nq, 40)]))
    for i in planted:
        d = int(rng.integers(1, 40))
        for _ in range(3):
            train[int(rng.integers(0, nt))] = _flip(rng, q[i], d)
    wi, wd, ties = o.match_mih(q, train)
    assert (ties[planted] > 1).sum() >= min(len(planted), 3) - 1
    gi, gd = fe.associate(q, train)
    assert np.array_equal(gd, wd)
    assert np.array_equal(gi, wi), "%d of %d indices differ" % ((gi != wi).sum(), nq)
    fe.close()
