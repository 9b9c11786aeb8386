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


def _drain_error(a):
    """True if the map had a failing update to report (lanefront.h: reported once, by the first call that sees it)."""
    try:
        a.state()
        return False
    except LanefrontError as e:
        assert e.code == -2
        a.state()                                            # ... and only once
        return True


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
    overflowed = pending = False
    reported = 0
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
            # an earlier update overflowed: reported ONCE, by a call that then did nothing -- the same call again works
            assert when_full == "error" and e.code == -2 and pending
            reported += 1
            pending = False
            sh.step(None, out, n, n_frames, idx, dd, poses=poses, step=step)
            a.synchronize()
        free = cap - o.state()["size"]
        appended = o.state()["total_appended"]
        oi, od = o.step(code, color, keep, ground, step, frame_offset=fo, poses=poses)
        if when_full == "error" and o.state()["total_appended"] - appended > free:
            overflowed, pending = True, True                 # this update dropped segments: the map owes one report
        assert np.array_equal(idx[:n].cpu().numpy(), oi) and np.array_equal(dd[:n].cpu().numpy(), od), step
        # the block this step exchanged, byte for byte against the numpy statement of the layout
        blk = sh.block.cpu().numpy()
        assert block_header(blk) == (n, step, n_frames)
        rows = block_rows(blk)
        assert np.array_equal(rows["code"], code) and np.array_equal(rows["color"], color) and np.array_equal(rows["keep"], keep)
        assert np.array_equal(rows["idx"], oi) and np.array_equal(rows["dist"], od)
        assert np.array_equal(rows["ground"], o.to_map_frame(ground, fo, poses)) and not rows["pad"].any()
        if pending and step % 2:
            assert _drain_error(a)                          # lf_map_size reports it as well (on the other steps: the next step call)
            reported += 1
            pending = False
        elif not pending:
            _same_map(a, o)                                 # the full map goes on being matched and refreshed
    if pending:
        assert _drain_error(a)
        reported += 1
    _same_map(a, o)
    assert overflowed == (when_full == "error") and (reported > 0) == overflowed
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


def _upload(torch, dev, c):
    return {"frame_offset": torch.from_numpy(c["frame_offset"]).to(dev), "code": torch.from_numpy(c["code"]).to(dev),
            "color": torch.from_numpy(c["color"]).to(dev), "keep": torch.from_numpy(c["keep"]).to(dev),
            "ground": torch.from_numpy(c["ground"]).to(dev)}


@pytest.mark.parametrize("n_blocks,when_full", [(2, "ring"), (8, "ring"), (8, "error"), (3, "ring")])
def test_multi_block_update_matches_oracle(n_blocks, when_full):
    """lf_map_update with SEVERAL blocks built on the device -- what every rank of a multi-GPU step applies after the
    all-gather (BASELINE configs[3]) -- against the oracle's update of the concatenated rows: ragged
    counts, an empty block, a completely full block, refreshes of ONE entry from different blocks, appends that wrap
    the ring inside the update (and, n_blocks = 3, more appends than the map holds in one update)."""
    import torch
    from oracle.oracle import OracleMap
    rng = np.random.default_rng(100 * n_blocks + len(when_full))
    G = 1500 if n_blocks == 3 else 200                    # rows of a block: the 3-block case crosses the 1024-row workgroups
    cap = 100 if when_full == "error" else (400 if n_blocks == 3 else 300)
    MD = 10
    kw = dict(capacity=cap, color_gating=True, max_distance=128, policy="merge", kept_only=True, merge_distance=MD,
              when_full=when_full)
    a, o = LineAssociator(**kw), OracleMap(**kw)
    dev = torch.device("cuda")
    pool = _codes(rng, 60)
    pool_color = rng.integers(0, 3, 60).astype(np.uint8)
    a.seed(pool[:30], pool_color[:30]); o.seed(pool[:30], pool_color[:30])
    rows = G + 1
    blocks = torch.zeros(n_blocks * rows * BLOCK_ROW_BYTES, dtype=torch.uint8, device=dev)
    saw_error = saw_shared = False
    for step in range(6):
        counts = [int(rng.integers(1, G + 1)) for _ in range(n_blocks)]
        if step % 2 == 0:
            counts[step % n_blocks] = 0                    # an empty block
        counts[(step + 1) % n_blocks] = G                  # a full one
        parts, keepalive = [], []
        torch.cuda.synchronize()
        for b in range(n_blocks):
            n = counts[b]
            src = rng.integers(0, 60, n)
            c = {"frame_offset": np.array([0, n // 3, n], np.int32), "code": _noisy(rng, pool[src], 12), "color": pool_color[src].copy(),
                 "keep": (rng.random(n) < 0.85).astype(np.uint8), "ground": rng.normal(size=(n, 4))}
            poses = rng.normal(size=(2, 3))
            out = _upload(torch, dev, c)
            idx = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
            dd = torch.zeros(max(n, 1), dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            ptrs = {k: v.data_ptr() for k, v in out.items()}
            if n:
                a.associate_device(None, ptrs["code"], ptrs["color"], n, idx.data_ptr(), dd.data_ptr())
            a.pack_block_device(None, ptrs, n, 2, idx.data_ptr(), dd.data_ptr(), poses, step,
                                blocks.data_ptr() + b * rows * BLOCK_ROW_BYTES, rows)
            keepalive.append((out, idx, dd))
            c["ground_map"] = o.to_map_frame(c["ground"], c["frame_offset"], poses)
            parts.append(c)
        a.update_device(blocks.data_ptr(), n_blocks, rows)
        a.synchronize()
        cat = {k: np.concatenate([p[k] for p in parts]) for k in ("code", "color", "keep", "ground_map")}
        oi, od = o.associate(cat["code"], cat["color"])
        # the gathered blocks carry every segment's association result against the map BEFORE the update
        raw = blocks.cpu().numpy().reshape(n_blocks, rows * BLOCK_ROW_BYTES)
        got = np.concatenate([block_rows(raw[b]) for b in range(n_blocks)])
        assert [block_header(raw[b])[0] for b in range(n_blocks)] == counts
        assert np.array_equal(got["idx"], oi) and np.array_equal(got["dist"], od), step
        assert np.array_equal(got["ground"], cat["ground_map"]) and np.array_equal(got["code"], cat["code"])
        free, appended = cap - o.state()["size"], o.state()["total_appended"]
        o.update(cat["code"], cat["color"], cat["keep"], cat["ground_map"], oi, od, step)
        dropped = when_full == "error" and o.state()["total_appended"] - appended > free
        assert _drain_error(a) == dropped                  # a failing update is reported, once; the map goes on
        saw_error |= dropped
        _same_map(a, o)
        # several segments of DIFFERENT blocks refreshed one entry in this update (the last in order must have won)
        ref = (oi >= 0) & (od <= MD) & (cat["keep"] != 0)
        tgt, cnt = np.unique(oi[ref], return_counts=True)
        blk_of = np.repeat(np.arange(n_blocks), counts)
        saw_shared |= any(len(set(blk_of[ref & (oi == t)])) > 1 for t in tgt[cnt > 1])
    st = o.state()
    assert st["total_refreshed"] > 0 and saw_error == (when_full == "error") and saw_shared
    if when_full == "ring":
        assert st["total_appended"] > cap
    # a block whose header is damaged: NOTHING of that update is applied, reported once, the map goes on
    before = a.fetch()
    bad = blocks.clone()
    bad[rows * BLOCK_ROW_BYTES] = 0                                           # block 1's magic
    torch.cuda.synchronize()
    a.update_device(bad.data_ptr(), n_blocks, rows)
    with pytest.raises(LanefrontError) as ei:
        a.state()
    assert ei.value.code == -1
    after = a.fetch()
    assert all(np.array_equal(before[k], after[k]) for k in before) and a.state()["size"] == st["size"]
    a.close()


def test_sharded_step_under_a_world_size_one_nccl_group(tmp_path):
    """The RCCL leg of the multi-GPU step on ONE GPU: init_process_group("nccl"), all_gather_into_tensor on the device
    blocks, the ExternalStream ordering against the map's stream -- ShardedAssociator(force_collective=True) must give
    what the no-collective path gives (and both what the oracle gives).  Also the collective failure: a batch that
    does not fit the block raises AFTER the gather, the replica skips the step and reports it once."""
    import torch
    import torch.distributed as dist
    from oracle.oracle import OracleMap
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29631")
    dist.init_process_group("nccl", rank=0, world_size=1, init_method="file://%s" % (tmp_path / "rdzv"))
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        kw = dict(capacity=256, color_gating=True, max_distance=128, policy="merge", kept_only=True, merge_distance=25)
        a_c, a_p, o = LineAssociator(**kw), LineAssociator(**kw), OracleMap(**kw)
        sh_c = ShardedAssociator(a_c, block_segments=300, device=dev, backend="nccl", force_collective=True)
        sh_p = ShardedAssociator(a_p, block_segments=300, device=dev, backend="nccl")
        assert sh_c.collective and not sh_p.collective
        rng = np.random.default_rng(5)
        pool = _codes(rng, 80)
        for step in range(7):
            n_frames = 4
            counts = rng.integers(0, 70, n_frames)
            n = int(counts.sum())
            c = {"frame_offset": np.concatenate([[0], np.cumsum(counts)]).astype(np.int32),
                 "code": _noisy(rng, pool[rng.integers(0, 80, n)], 16), "color": rng.integers(0, 3, n).astype(np.uint8),
                 "keep": (rng.random(n) < 0.8).astype(np.uint8), "ground": rng.normal(size=(n, 4))}
            poses = rng.normal(size=(n_frames, 3))
            out = _upload(torch, dev, c)
            res = []
            for sh in (sh_c, sh_p):
                idx = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
                dd = torch.zeros(max(n, 1), dtype=torch.float32, device=dev)
                torch.cuda.synchronize()
                sh.step(None, out, n, n_frames, idx, dd, poses=poses, step=step)
                sh.map.synchronize()
                torch.cuda.synchronize()
                res.append((idx[:n].cpu().numpy(), dd[:n].cpu().numpy()))
            oi, od = o.step(c["code"], c["color"], c["keep"], c["ground"], step, frame_offset=c["frame_offset"], poses=poses)
            for gi, gd in res:
                assert np.array_equal(gi, oi) and np.array_equal(gd, od), step
            assert np.array_equal(sh_c.gathered.cpu().numpy(), sh_p.block.cpu().numpy())
            _same_map(a_c, o)
            _same_map(a_p, o)
        assert o.state()["total_refreshed"] > 0 and o.state()["total_appended"] > 256
        # collective failure
        n = 400
        c = {"frame_offset": np.array([0, n], np.int32), "code": _codes(rng, n), "color": np.zeros(n, np.uint8),
             "keep": np.ones(n, np.uint8), "ground": np.zeros((n, 4))}
        out = _upload(torch, dev, c)
        idx = torch.zeros(n, dtype=torch.int32, device=dev)
        dd = torch.zeros(n, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        with pytest.raises(LanefrontError) as ei:
            sh_c.step(None, out, n, 1, idx, dd, step=7)
        assert ei.value.code == -2
        hdr = np.frombuffer(sh_c.gathered[:BLOCK_ROW_BYTES].cpu().numpy().tobytes(), "<u4")
        assert hdr[1] == 0 and hdr[4] == n                                    # header only, overflow marker
        with pytest.raises(LanefrontError) as ei:
            a_c.state()                                                       # the replica skipped the step: reported once
        assert ei.value.code == -2
        _same_map(a_c, o)
        a_c.close(); a_p.close()
    finally:
        dist.destroy_process_group()


def _shared_gpu_rank(rank, world, port, q):
    """One rank of the two-process shared-GPU test (spawned: a fresh process, nothing exec'ed after HIP is up)."""
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from test_distributed_gloo import STEPS, _chunk
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    a = LineAssociator(**SHARED_KW)
    sh = ShardedAssociator(a, block_segments=32, device=dev, backend="gloo")
    res = []
    for step in range(STEPS):
        c = _chunk(step, rank)
        out = _upload(torch, dev, c)
        idx = torch.zeros(max(c["n"], 1), dtype=torch.int32, device=dev)
        dd = torch.zeros(max(c["n"], 1), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        sh.step(None, out, c["n"], 2, idx, dd, poses=c["poses"], step=step)
        a.synchronize()
        res.append((idx[:c["n"]].cpu().numpy(), dd[:c["n"]].cpu().numpy()))
    q.put((rank, res, a.fetch(), a.state()))
    dist.barrier()
    a.close()
    dist.destroy_process_group()


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHARED_KW = dict(capacity=64, color_gating=True, max_distance=128, policy="merge", kept_only=True, merge_distance=30, when_full="ring")


def test_two_processes_sharing_the_gpu_hold_identical_replicas():
    """The GPU twin of tests/test_distributed_gloo.py: two spawned processes share device 0, each with a REAL device
    map, blocks staged through gloo -- the replicas must be identical, equal to the one-rank device map over the
    concatenated segments, and equal to the oracle's."""
    import sys
    import torch
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle.oracle import OracleMap
    from test_distributed_gloo import STEPS, _chunk, _free_port, _merge_chunks
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shared_gpu_rank, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for k in ("code", "color", "ground", "hits", "last_seen"):
        assert np.array_equal(res[0][2][k], res[1][2][k]), k
    assert res[0][3] == res[1][3]
    dev = torch.device("cuda:0")
    a, o = LineAssociator(**SHARED_KW), OracleMap(**SHARED_KW)
    one = ShardedAssociator(a, block_segments=64, device=dev, backend="gloo")
    for step in range(STEPS):
        c = _merge_chunks(step, world)
        out = _upload(torch, dev, c)
        idx = torch.zeros(max(c["n"], 1), dtype=torch.int32, device=dev)
        dd = torch.zeros(max(c["n"], 1), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        one.step(None, out, c["n"], 4, idx, dd, poses=c["poses"], step=step)
        a.synchronize()
        oi, od = o.step(c["code"], c["color"], c["keep"], c["ground"], step, frame_offset=c["frame_offset"], poses=c["poses"])
        gi, gd = idx[:c["n"]].cpu().numpy(), dd[:c["n"]].cpu().numpy()
        assert np.array_equal(gi, oi) and np.array_equal(gd, od)
        n0 = _chunk(step, 0)["n"]
        for r, sl in ((0, slice(0, n0)), (1, slice(n0, None))):
            assert np.array_equal(gi[sl], res[r][1][step][0]) and np.array_equal(gd[sl], res[r][1][step][1])
    _same_map(a, o)
    assert a.state() == res[0][3]
    mine = a.fetch()
    for k in ("code", "color", "ground", "hits", "last_seen"):
        assert np.array_equal(mine[k], res[0][2][k]), k
    assert o.state()["total_refreshed"] > 0 and o.state()["total_appended"] > SHARED_KW["capacity"]
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
