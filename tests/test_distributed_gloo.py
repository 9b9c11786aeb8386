"""world_size-2 CPU test (gloo) of the multi-GPU step protocol, lane_slam_amd.distributed.ShardedAssociator -- the
very class bench.py drives on MI355Xs -- with the oracle's live map standing in for the device map
(tests/host_map.py).  SURVEY 8(e): with frames dealt to ranks in contiguous chunks, the replicas of a 2-rank run and
the map of a 1-rank run over the same segments are identical entry for entry, and so are the matches."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = 6
MAP_KW = dict(capacity=48, color_gating=True, max_distance=128, policy="merge", kept_only=True, merge_distance=30,
              when_full="ring")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _chunk(step, rank):
    """Segments of `rank`'s frames at `step` (two frames per rank, ragged counts): deterministic in (step, rank)."""
    rng = np.random.default_rng(1000 * step + rank)
    pool = np.random.default_rng(7).integers(0, 256, (40, 32), dtype=np.uint8)      # recurring lane markings
    counts = [int(rng.integers(0, 9)), int(rng.integers(3, 12))]
    n = sum(counts)
    code = pool[rng.integers(0, 40, n)].copy()
    for i in range(n):                                                                 # a few bits of noise each
        for b in rng.choice(256, size=int(rng.integers(0, 12)), replace=False):
            code[i, b >> 3] ^= np.uint8(1 << (b & 7))
    return {"frame_offset": np.array([0, counts[0], n], np.int32), "code": code,
            "color": rng.integers(0, 3, n).astype(np.uint8), "keep": (rng.random(n) < 0.8).astype(np.uint8),
            "ground": rng.normal(size=(n, 4)), "poses": rng.normal(size=(2, 3)), "n": n}


def _run(sharded, chunks_of_step):
    """Drive ShardedAssociator.step over STEPS steps; returns the per-step (idx, dist) and the final map."""
    res = []
    for step in range(STEPS):
        c = chunks_of_step(step)
        out = {k: torch.from_numpy(c[k]) for k in ("frame_offset", "code", "color", "keep", "ground")}
        idx = torch.zeros(max(c["n"], 1), dtype=torch.int32)
        dd = torch.zeros(max(c["n"], 1), dtype=torch.float32)
        sharded.step(None, out, c["n"], len(c["frame_offset"]) - 1, idx, dd, poses=c["poses"], step=step)
        res.append((idx[:c["n"]].numpy().copy(), dd[:c["n"]].numpy().copy()))
    return res, sharded.map.o.fetch(), sharded.map.o.state()


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from host_map import HostMap
    from lane_slam_amd.distributed import ShardedAssociator
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = ShardedAssociator(HostMap(**MAP_KW), block_segments=32, device="cpu", backend="gloo")
    res, fetched, state = _run(sh, lambda step: _chunk(step, rank))
    q.put((rank, res, fetched, state))
    dist.barrier()
    dist.destroy_process_group()


def _merge_chunks(step, world):
    cs = [_chunk(step, r) for r in range(world)]
    fo = [0]
    for c in cs:
        for v in np.diff(c["frame_offset"]):
            fo.append(fo[-1] + int(v))
    out = {k: np.concatenate([c[k] for c in cs]) for k in ("code", "color", "keep", "ground", "poses")}
    out["frame_offset"] = np.array(fo, np.int32)
    out["n"] = sum(c["n"] for c in cs)
    return out


def test_two_rank_replicas_equal_the_one_rank_map():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # replicas identical
    for k in ("code", "color", "ground", "hits", "last_seen"):
        assert np.array_equal(res[0][2][k], res[1][2][k]), k
    assert res[0][3] == res[1][3]
    # the 1-rank run: the same segments, each step's chunks concatenated in rank order (= frames in contiguous
    # chunks), through the same ShardedAssociator code with no process group
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from host_map import HostMap
    from lane_slam_amd.distributed import ShardedAssociator
    one = ShardedAssociator(HostMap(**MAP_KW), block_segments=64, device="cpu", backend="gloo")
    res1, fetched1, state1 = _run(one, lambda step: _merge_chunks(step, world))
    assert state1 == res[0][3]
    assert state1["total_refreshed"] > 0 and state1["total_appended"] > MAP_KW["capacity"]     # merges happened, the ring wrapped
    for k in ("code", "color", "ground", "hits", "last_seen"):
        assert np.array_equal(fetched1[k], res[0][2][k]), k
    for step in range(STEPS):
        n0 = _chunk(step, 0)["n"]
        for r, sl in ((0, slice(0, n0)), (1, slice(n0, None))):
            assert np.array_equal(res1[step][0][sl], res[r][1][step][0])      # idx
            assert np.array_equal(res1[step][1][sl], res[r][1][step][1])      # dist


def _big_chunk(step, rank, n):
    rng = np.random.default_rng(77 * step + rank)
    return {"frame_offset": np.array([0, n // 2, n], np.int32), "code": rng.integers(0, 256, (n, 32), dtype=np.uint8),
            "color": rng.integers(0, 3, n).astype(np.uint8), "keep": np.ones(n, np.uint8), "ground": rng.normal(size=(n, 4)),
            "poses": rng.normal(size=(2, 3)), "n": n}


def _overflow_worker(rank, world, port, q):
    """Step 1: rank 1 alone has more segments than a block holds.  Both ranks must come out of the step with an
    error (no rank left waiting in the all-gather), hold unchanged and identical replicas, and go on."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from host_map import HostMap
    from lane_slam_amd.distributed import ShardedAssociator
    from lane_slam_amd.frontend import LanefrontError
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = ShardedAssociator(HostMap(**MAP_KW), block_segments=32, device="cpu", backend="gloo")
    raised, sizes = [], []
    for step in range(4):
        c = _big_chunk(step, rank, 40) if (step == 1 and rank == 1) else _chunk(step, rank)
        out = {k: torch.from_numpy(c[k]) for k in ("frame_offset", "code", "color", "keep", "ground")}
        idx = torch.zeros(max(c["n"], 1), dtype=torch.int32)
        dd = torch.zeros(max(c["n"], 1), dtype=torch.float32)
        try:
            sh.step(None, out, c["n"], 2, idx, dd, poses=c["poses"], step=step)
        except LanefrontError as e:
            raised.append((step, e.code))
        sizes.append(sh.map.o.state()["total_appended"])
    q.put((rank, raised, sizes, sh.map.o.fetch(), sh.map.o.state()))
    dist.barrier()
    dist.destroy_process_group()


def test_one_rank_overflowing_fails_the_step_on_every_rank():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overflow_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])        # a hang would time out here
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        assert res[r][1] == [(1, -2)], res[r][1]                   # LF_ERR_CAPACITY at step 1, on both ranks, nowhere else
        assert res[r][2][1] == res[r][2][0] and res[r][2][2] > res[r][2][1]          # step 1 applied nothing; step 2 went on
    for k in ("code", "color", "ground", "hits", "last_seen"):
        assert np.array_equal(res[0][3][k], res[1][3][k]), k
    assert res[0][4] == res[1][4]
    # = the one-rank map over steps 0, 2, 3
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from host_map import HostMap
    from lane_slam_amd.distributed import ShardedAssociator
    one = ShardedAssociator(HostMap(**MAP_KW), block_segments=64, device="cpu", backend="gloo")
    for step in (0, 2, 3):
        c = _merge_chunks(step, world)
        out = {k: torch.from_numpy(c[k]) for k in ("frame_offset", "code", "color", "keep", "ground")}
        one.step(None, out, c["n"], 4, torch.zeros(max(c["n"], 1), dtype=torch.int32), torch.zeros(max(c["n"], 1), dtype=torch.float32),
                 poses=c["poses"], step=step)
    assert one.map.o.state() == res[0][4]
    f1 = one.map.o.fetch()
    for k in ("code", "color", "ground", "hits", "last_seen"):
        assert np.array_equal(f1[k], res[0][3][k]), k


def test_block_overflow_is_an_error_not_a_truncation():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from host_map import HostMap
    from lane_slam_amd.distributed import ShardedAssociator
    sh = ShardedAssociator(HostMap(**MAP_KW), block_segments=4, device="cpu", backend="gloo")
    c = _merge_chunks(0, 2)
    assert c["n"] > 4
    out = {k: torch.from_numpy(c[k]) for k in ("frame_offset", "code", "color", "keep", "ground")}
    with pytest.raises(RuntimeError):
        sh.step(None, out, c["n"], len(c["frame_offset"]) - 1, torch.zeros(c["n"], dtype=torch.int32),
                torch.zeros(c["n"], dtype=torch.float32), poses=c["poses"], step=0)
    assert sh.map.o.state()["size"] == 0
