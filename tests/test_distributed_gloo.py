"""world_size-2 CPU test (gloo) of the multi-GPU merge path: the all-gather of padded segment
blocks must give every rank the same merged list in rank-major order, equal to what a single
rank would have produced."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from lane_slam_amd.distributed import LiveMap, all_gather_blocks, merged_codes, pack_block
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(100 + rank)
    n = 5 + 3 * rank                                       # ragged: ranks produce different counts
    code = torch.from_numpy(rng.integers(0, 256, (n, 32), dtype=np.uint8))
    keep = torch.from_numpy((rng.random(n) < 0.7).astype(np.uint8))
    color = torch.from_numpy(rng.integers(0, 3, n).astype(np.uint8))
    block = torch.zeros(16, 34, dtype=torch.uint8)
    k = pack_block(block, code, keep, color, n)
    gathered, counts = all_gather_blocks(block, k)
    merged = merged_codes(gathered, counts, kept_only=True)
    m = LiveMap(12)
    m.append(merged)
    q.put((rank, counts.tolist(), merged.numpy().copy(), m.view().numpy().copy(),
           code.numpy(), keep.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_merge_is_identical_on_every_rank():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == [5, 8]
    assert np.array_equal(res[0][2], res[1][2])            # same merged list everywhere
    assert np.array_equal(res[0][3], res[1][3])            # same map replica everywhere
    # rank-major, segment-minor, kept only == what one rank holding both shards would append
    expect = np.concatenate([res[r][4][res[r][5] != 0] for r in range(world)])
    assert np.array_equal(res[0][2], expect)


def test_live_map_wraps():
    sys.path.insert(0, ROOT)
    from lane_slam_amd.distributed import LiveMap
    m = LiveMap(5)
    a = torch.arange(4 * 32, dtype=torch.uint8).reshape(4, 32)
    m.append(a)
    assert m.size == 4
    m.append(a[:3] + 100)
    assert m.size == 5 and m.head == 2
    assert torch.equal(m.codes[4], a[0] + 100) and torch.equal(m.codes[0], a[1] + 100)
    m.append(torch.zeros(9, 32, dtype=torch.uint8))
    assert m.size == 5 and int(m.codes.sum()) == 0
