"""GPU parity tests of the live map + associator (lf_map_*, k_assoc.hip / k_map.hip) against the oracle's sequential
statement of the same contract (oracle/lf_oracle_map.c), and the BASELINE configs[2] streaming test: >= 1000 distinct
frames through detect -> describe -> project -> sanity -> associate -> map update, compared with the oracle run in
the same loop, step by step."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from lane_slam_amd import FrontEnd, LanefrontError, LineAssociator, default_config, synth
from lane_slam_amd.distributed import BLOCK_ROW_BYTES, ShardedAssociator, block_header, block_rows

pytestmark = pytest.mark.gpu


def _codes(rng, n):
    return rng.integers(0, 256, (n, 32), dtype=np.uint8)


def _noisy(rng, src, max_bits):
    out = src.copy()
    for i in range(out.shape[0]):
        for b in rng.choice(256, size=int(rng.integers(0, max_bits + 1)), replace=False):
            out[i, b >> 3] ^= np.uint8(1 << (b & 7))
    return out


def _same_map(a, o, upto=None):
    st_g, st_o = a.state(), o.state()
    for k in ("size", "head", "total_appended", "total_refreshed"):
        assert st_g[k] == st_o[k], (k, st_g, st_o)
    g, r = a.fetch(), o.fetch()
    n = st_o["size"] if upto is None else upto
    for k in ("code", "color", "ground", "hits", "last_seen"):
        assert np.array_equal(g[k][:n], r[k][:n]), k


@pytest.mark.parametrize("gating", [False, True])
def test_map_association_matches_oracle(gating):
    from oracle.oracle import OracleMap
    rng = np.random.default_rng(11 + gating)
    kw = dict(capacity=40000, color_gating=gating, max_distance=128, kept_only=False)
    a, o = LineAssociator(**kw), OracleMap(**kw)
    # empty map
    q = _codes(rng, 9)
    qc = rng.integers(0, 3, 9).astype(np.uint8)
    i0, d0 = a.associate(q, qc)
    assert (i0 == -1).all() and (d0 == -1).all()
    sizes = [1, 62, 1, 64, 3000, 14000, 17000]            # map sizes cross the 64-row tile and the 16384-column chunk
    total = 0
    for k, n in enumerate(sizes):
        m = _codes(rng, n)
        mc = rng.integers(0, 3, n).astype(np.uint8)
        if k == 4:
            mc[::7] = 255                                   # wildcards
        a.seed(m, mc)
        o.seed(m, mc)
        total += n
        nq = [5, 257, 700][k % 3]
        src = a.fetch(0, total)["code"]
        q = _noisy(rng, src[rng.integers(0, total, nq)], 150)       # from exact copies to beyond 128 bits
        q[0] = src[total - 1]                                        # the newest entry, exactly
        qc = rng.integers(0, 3, nq).astype(np.uint8)
        qc[1 % nq] = 255
        gi, gd = a.associate(q, qc)
        oi, od = o.associate(q, qc)
        assert np.array_equal(gd, od), (k, n)
        assert np.array_equal(gi, oi), (k, n)                       # both sides resolve ties to the lowest index
        if not gating:
            assert gi[0] >= 0 and gd[0] == 0
    _same_map(a, o)
    # a smaller cut
    a2, o2 = LineAssociator(capacity=4096, max_distance=20, kept_only=False), OracleMap(capacity=4096, max_distance=20, kept_only=False)
    m = _codes(rng, 1000)
    a2.seed(m); o2.seed(m)
    q = _noisy(rng, m[:400], 40)
    gi, gd = a2.associate(q)
    oi, od = o2.associate(q)
    assert np.array_equal(gi, oi) and np.array_equal(gd, od) and (gi == -1).any() and (gi >= 0).any()
    with pytest.raises(LanefrontError):
        LineAssociator(capacity=4096, color_gating=True).associate(q)         # gating needs colours
    for x in (a, a2):
        x.close()


@pytest.mark.parametrize("policy,when_full,cap", [("append", "ring", 700), ("merge", "ring", 200), ("merge", "error", 256),
                                                  ("append", "ring", 64)])
def test_map_update_matches_oracle(policy, when_full, cap):
    import torch
    from oracle.oracle import OracleMap
    rng = np.random.default_rng(sum(map(ord, policy + when_full)) + cap)
    kw = dict(capacity=cap, color_gating=True, max_distance=128, policy=policy, kept_only=True, merge_distance=25,
              when_full=when_full)
    a, o = LineAssociator(**kw), OracleMap(**kw)
    sh = ShardedAssociator(a, block_segments=600, device="cuda")
    pool = _codes(rng, 120)
    dev = torch.device("cuda")
    overflowed = False
    for step in range(9):
        n_frames = 5
        counts = rng.integers(0, 90, n_frames)
        if step == 3:
            counts[:] = 0                                   # an empty batch
        n = int(counts.sum())
        fo = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        code = _noisy(rng, pool[rng.integers(0, 120, n)], 18) if n else np.zeros((0, 32), np.uint8)
        color = rng.integers(0, 3, n).astype(np.uint8)
        keep = (rng.random(n) < 0.8).astype(np.uint8)
        ground = rng.normal(size=(n, 4))
        poses = np.column_stack([rng.normal(size=n_frames), rng.normal(size=n_frames), rng.uniform(-3.2, 3.2, n_frames)])
        out = {"frame_offset": torch.from_numpy(fo).to(dev), "code": torch.from_numpy(code).to(dev),
               "color": torch.from_numpy(color).to(dev), "keep": torch.from_numpy(keep).to(dev),
               "ground": torch.from_numpy(ground).to(dev)}
        idx = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
        dd = torch.zeros(max(n, 1), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        try:
            sh.step(None, out, n, n_frames, idx, dd, poses=poses, step=step)
            a.synchronize()
        except LanefrontError as e:
            assert when_full == "error" and e.code == -2     # the previous update overflowed: reported at the next call
            overflowed = True
            break
        oi, od = o.step(code, color, keep, ground, step, frame_offset=fo, poses=poses)
        assert np.array_equal(idx[:n].cpu().numpy(), oi) and np.array_equal(dd[:n].cpu().numpy(), od), step
        # the block this step exchanged, byte for byte against the numpy statement of the layout
        blk = sh.block.cpu().numpy()
        assert block_header(blk) == (n, step, n_frames)
        rows = block_rows(blk)
        assert np.array_equal(rows["code"], code) and np.array_equal(rows["color"], color) and np.array_equal(rows["keep"], keep)
        assert np.array_equal(rows["idx"], oi) and np.array_equal(rows["dist"], od)
        assert np.array_equal(rows["ground"], o.to_map_frame(ground, fo, poses)) and not rows["pad"].any()
        if o.state()["overflow"]:
            with pytest.raises(LanefrontError):
                a.state()
            overflowed = True
            break
        _same_map(a, o)
    assert overflowed == (when_full == "error")
    if policy == "merge":
        assert o.state()["total_refreshed"] > 0
    if when_full == "ring":
        assert o.state()["total_appended"] > cap             # the ring wrapped (cap 64: more appends than capacity in one update)
    # a block too small for the batch is an error, never a truncation
    small = ShardedAssociator(a, block_segments=3, device="cuda") if when_full == "ring" else None
    if small is not None:
        n = 10
        out = {"frame_offset": torch.tensor([0, n], dtype=torch.int32, device=dev), "code": torch.zeros(n, 32, dtype=torch.uint8, device=dev),
               "color": torch.zeros(n, dtype=torch.uint8, device=dev), "keep": torch.ones(n, dtype=torch.uint8, device=dev),
               "ground": torch.zeros(n, 4, dtype=torch.float64, device=dev)}
        before = a.state()
        with pytest.raises(LanefrontError):
            small.step(None, out, n, 1, torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.float32, device=dev))
        assert a.state() == before
    a.close()


def _oracle_frames(cfg, frames, threads):
    from oracle.oracle import Oracle
    oracles = [Oracle(cfg) for _ in range(threads)]

    def work(t):
        return [(f, oracles[t].process_frame(frames[f], cap=3 * 512)) for f in range(t, frames.shape[0], threads)]
    with ThreadPoolExecutor(threads) as ex:
        res = dict(kv for part in ex.map(work, range(threads)) for kv in part)
    return [res[f] for f in range(frames.shape[0])]


@pytest.mark.parametrize("mode", ["append", "merge_gated"])
def test_stream_replay_matches_oracle_loop(mode):
    """BASELINE configs[2]: a 1024-frame stream of distinct synthetic 640x480 frames, full-res
    geometry, through the whole path in 128-frame batches, against a map that grows by the kept segments
    (show_map.py:28-42), with per-frame odometry poses.  Every step's idx / dist and the final map must equal the
    oracle's, which runs the same loop (frames -> oracle front end -> oracle map)."""
    import torch
    from oracle.oracle import OracleMap
    cfg = default_config("fullres")
    B, steps = 128, 8
    dev = torch.device("cuda")
    if mode == "append":
        kw = dict(capacity=65536, color_gating=False, max_distance=128, policy="append", kept_only=True)
    else:
        kw = dict(capacity=2048, color_gating=True, max_distance=128, policy="merge", kept_only=True, merge_distance=30)
    a, o = LineAssociator(**kw), OracleMap(**kw)
    fe = FrontEnd(cfg, max_frames=B, max_lines_per_color=512)
    cap = B * 3 * 512
    out = {"frame_offset": torch.zeros(B + 1, dtype=torch.int32, device=dev), "lines": torch.zeros(cap, 4, dtype=torch.float32, device=dev),
           "normals": torch.zeros(cap, 2, dtype=torch.float32, device=dev), "color": torch.zeros(cap, dtype=torch.uint8, device=dev),
           "pixels_normalized": torch.zeros(cap, 4, dtype=torch.float32, device=dev), "ground": torch.zeros(cap, 4, dtype=torch.float64, device=dev),
           "keep": torch.zeros(cap, dtype=torch.uint8, device=dev), "desc": torch.zeros(cap, 72, dtype=torch.float32, device=dev),
           "code": torch.zeros(cap, 32, dtype=torch.uint8, device=dev)}
    ptrs = {k: v.data_ptr() for k, v in out.items()}
    idx = torch.zeros(cap, dtype=torch.int32, device=dev)
    dd = torch.zeros(cap, dtype=torch.float32, device=dev)
    threads = max(1, min(32, (os.cpu_count() or 2) // 2))
    seen, matched = 0, 0
    for step in range(steps):
        frames = synth.make_batch(B, seed0=50000 + step * B, threads=threads)                   # 1024 distinct frames in all
        t = np.arange(step * B, (step + 1) * B, dtype=np.float64)
        poses = np.column_stack([0.01 * t, 0.3 * np.sin(0.02 * t), 0.002 * t])   # a gentle left-hand arc
        d_frames = torch.from_numpy(frames).to(dev)
        torch.cuda.synchronize()
        fe.submit_device(d_frames.data_ptr(), B, ptrs, cap, describe=True)
        n = fe.wait()
        a.step_device(fe, ptrs, n, B, idx.data_ptr(), dd.data_ptr(), poses=poses, step=step)
        a.synchronize()
        ref = _oracle_frames(cfg, frames, threads)
        r = {k: np.concatenate([x[k] for x in ref]) for k in ("code", "color", "keep", "ground")}
        fo = np.concatenate([[0], np.cumsum([x["n"] for x in ref])]).astype(np.int32)
        assert n == fo[-1] and np.array_equal(out["frame_offset"].cpu().numpy(), fo)
        assert np.array_equal(out["code"][:n].cpu().numpy(), r["code"]) and np.array_equal(out["keep"][:n].cpu().numpy(), r["keep"])
        oi, od = o.step(r["code"], r["color"], r["keep"], r["ground"], step, frame_offset=fo, poses=poses)
        gi, gd = idx[:n].cpu().numpy(), dd[:n].cpu().numpy()
        assert np.array_equal(gd, od), step                # distances: exact
        assert np.array_equal(gi, oi), step                # indices: both sides take the lowest index among equals
        seen += n
        matched += int((oi >= 0).sum())
    _same_map(a, o)
    st = o.state()
    assert seen > 20000 and matched > 0 and st["size"] > 1000
    if mode == "merge_gated":
        assert st["total_refreshed"] > 0
    print("stream %s: %d segments, %d matched, map %r" % (mode, seen, matched, st))
    fe.close()
    a.close()
