#!/usr/bin/env python3
"""Headline benchmark: frames/s through detect -> describe -> ground-project -> sanity ->
associate on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the whole hot path over one batch of 256 synthetic 640x480 BGR frames
per GPU, already resident in HBM (BASELINE.json configs[1], full-res geometry: img_size
[480,640], top_cutoff 160 -> 640x320 working image), followed by association of the step's
descriptors against a live map of 50 000 codes.  With N > 1 GPUs frames shard across ranks
(weak scaling, 256 frames per GPU per step) and the per-rank segment blocks are merged with
one RCCL all-gather before the map is updated; association itself needs no collective (the
map is replicated, queries stay local).

Rank 0 prints ONE JSON line.  `roofline` describes the dominant streaming kernel with live
HIP-event timings from the library's own stream; `cpu_baseline` times the CPU oracle
(oracle/, a single-threaded restatement of the reference path) on a bounded sample of the
same workload on the GPU box's host.
"""
import argparse
import json
import os
import sys
import time

# Six batches are kept in flight on six HIP streams (plus torch's); ROCclr maps streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share a queue serialise.  Must be
# set before the HIP runtime initialises.  16 leaves room for the collective library's own streams at N > 1.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
# dmabuf IPC (the only kind the pool's host driver supports) -- read when HSA initialises, so it is set here, before
# any torch GPU call, not next to init_process_group
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

INT8_MFMA_PEAK_POPS = 5.0     # dense int8 MFMA, MI355X_MICROARCH.md (2x the ~2.5 PF bf16 rate)
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)


def stage_bytes(P, Ps, resize):
    """ALGORITHMIC HBM bytes per frame of each streaming kernel.  The LSD stages are
    not listed: their traffic is proportional to the number of edge pixels (sparse records), they are
    latency / issue bound and are reported as time only."""
    return {
        "pre(resize+correct+hsv+masks+dilate)": 3 * P + 3 * P + 3 * P + 3 * (P // 8),
        "canny_nms": 3 * P + 2 * (P // 8),
        "canny_hysteresis": 3 * (P // 8),
        "lbd_gray_blur_sobel": 3 * P + 4 * P,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--batch", type=int, default=0, help="frames per GPU per step (0: 256, or 128 for --geometry hd)")
    ap.add_argument("--geometry", default="fullres", choices=["fullres", "parity", "hd"],
                    help="fullres / parity: BASELINE configs[1] (640x480 input); hd: configs[4]'s 1920x1080 frames "
                         "(img_size [1080,1920], top_cutoff 360) -- an optional stress mode, not the headline")
    ap.add_argument("--map", type=int, default=50000, help="live-map size (codes)")
    ap.add_argument("--unique", type=int, default=64, help="distinct synthetic frames per rank (tiled to --batch)")
    ap.add_argument("--cap", type=int, default=512, help="max lines per (frame, colour)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo (host staging) only to dry-run the N>1 path on a shared GPU")
    ap.add_argument("--depth", type=int, default=0, help="independent batches in flight (handles/streams); 0: 6, or 3 for --geometry hd")
    ap.add_argument("--lsd-refine", type=int, default=2, help="diagnostic only: 0/1 skip refine / NFA stages (invalid as a headline run)")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames for the CPU baseline (0 = auto, -1 = skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from lane_slam_amd import FrontEnd, default_config, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("LF_SHARED_GPU"):          # dry runs: every rank on device 0
        local_rank = 0
    if world != args.gpus:
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # LF_FORCE_COLLECTIVES=1 runs the N > 1 code path (process group, all-gathers, rank-major merge) with a single
    # rank: a dry run of the RCCL calls on one GPU, never a headline configuration
    multi = world > 1 or bool(os.environ.get("LF_FORCE_COLLECTIVES"))
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    hd = args.geometry == "hd"
    B = args.batch or (128 if hd else 256)
    D = args.depth if args.depth > 0 else (3 if hd else 6)
    in_rows, in_cols = (1080, 1920) if hd else (480, 640)
    cfg = default_config("fullres", in_size=(in_rows, in_cols)) if hd else default_config(args.geometry)
    cfg["lsd"]["refine"] = args.lsd_refine
    # D handles = D independent batches in flight (each handle owns a HIP stream and its buffers)
    fes = [FrontEnd(cfg, device=local_rank, max_frames=B, max_lines_per_color=args.cap) for _ in range(D)]
    fe = fes[0]
    P, Ps = fe.rows * fe.cols, fe.lsd_rows * fe.lsd_cols

    # ---- synthetic input, resident in HBM before the timed region
    uniq = min(args.unique, B)
    host = synth.make_batch(uniq, seed0=10000 * rank, rows=in_rows, cols=in_cols)
    reps = (B + uniq - 1) // uniq
    host = np.ascontiguousarray(np.tile(host, (reps, 1, 1, 1))[:B])
    frames = torch.from_numpy(host).to(dev)

    cap = B * 3 * args.cap

    def alloc_out():
        return {
            "frame_offset": torch.zeros(B + 1, dtype=torch.int32, device=dev),
            "lines": torch.zeros(cap, 4, dtype=torch.float32, device=dev),
            "normals": torch.zeros(cap, 2, dtype=torch.float32, device=dev),
            "color": torch.zeros(cap, dtype=torch.uint8, device=dev),
            "pixels_normalized": torch.zeros(cap, 4, dtype=torch.float32, device=dev),
            "ground": torch.zeros(cap, 4, dtype=torch.float64, device=dev),
            "keep": torch.zeros(cap, dtype=torch.uint8, device=dev),
            "desc": torch.zeros(cap, 72, dtype=torch.float32, device=dev),
            "code": torch.zeros(cap, 32, dtype=torch.uint8, device=dev),
        }

    outs = [alloc_out() for _ in range(D)]
    ptrs = [{k: v.data_ptr() for k, v in o.items()} for o in outs]
    # live map: M random codes (seed 1234, identical on every rank) + a rolling region that
    # receives the segments all ranks produced (append-only map, show_map.py:28-42)
    G = 16 * 1024                                   # gathered segments per rank (fixed-capacity block; a step's
                                                    # ~11 k segments fit, a longer list contributes its first G)
    # The rolling region has a FIXED total size: every rank contributes roll / world of its newest segments per
    # step, so the map -- and with it the association work per GPU -- is the same at every N (weak scaling).
    roll = (min(G, 16384) // world) * world
    map_codes = torch.from_numpy(np.concatenate([synth.random_codes(args.map, 1234),
                                                 synth.random_codes(roll, 4321)])).to(dev)
    M = map_codes.shape[0]
    a_idx = [torch.zeros(cap, dtype=torch.int32, device=dev) for _ in range(D)]
    a_dist = [torch.zeros(cap, dtype=torch.float32, device=dev) for _ in range(D)]
    block = torch.zeros(G, 34, dtype=torch.uint8, device=dev)          # code(32) + keep + colour per segment
    gathered = torch.zeros(world * G, 34, dtype=torch.uint8, device=dev)
    counts_local = torch.zeros(1, dtype=torch.int32, device=dev)
    counts_all = torch.zeros(world, dtype=torch.int32, device=dev)
    seg_total = [0]
    # the handles' own HIP streams, wrapped so that torch events can order work across them
    exts = [torch.cuda.ExternalStream(f.stream_ptr(), device=dev) for f in fes]
    events = [tuple(torch.cuda.Event() for _ in range(3)) for _ in range(D)]
    host_ms = {"wait": 0.0, "merge": 0.0, "assoc": 0.0, "map": 0.0, "n": 0}

    def finish(slot):
        """Complete the batch queued on `slot`: merge segment lists across ranks, associate, update the map."""
        f, out = fes[slot], outs[slot]
        h0 = time.perf_counter()
        total = f.wait()
        h1 = time.perf_counter()
        seg_total[0] = total
        n = min(total, G)
        if multi:
            block[:n, :32] = out["code"][:n]
            block[:n, 32] = out["keep"][:n]
            block[:n, 33] = out["color"][:n]
            counts_local[0] = n
            if args.backend == "nccl":           # RCCL over xGMI, device buffers
                dist.all_gather_into_tensor(counts_all, counts_local)
                dist.all_gather_into_tensor(gathered, block)
            else:                                # gloo dry run: stage through the host
                hc, hb = counts_all.cpu(), gathered.cpu()
                dist.all_gather_into_tensor(hc, counts_local.cpu())
                dist.all_gather_into_tensor(hb, block.cpu())
                counts_all.copy_(hc)
                gathered.copy_(hb)
            src = gathered.view(world, G, 34)[:, : roll // world, :32]
        else:
            src = out["code"][: roll].view(1, -1, 32)[:, : roll]
        # Association: this rank's segments against the replicated map as it stood before this batch; then the map
        # update.  The association runs on the handle's own stream, the merge and the map on torch's: they are
        # ordered with events, not host synchronisation, so the host goes straight back to queueing the next batch
        # (with host syncs here the freed handle sat idle for ~1.4 ms per step behind a saturated GPU).
        cur = torch.cuda.current_stream()
        ev_a, ev_b, ev_c = events[slot]
        ev_a.record(cur)
        exts[slot].wait_event(ev_a)                 # after the merge and every earlier map update
        h2 = time.perf_counter()
        if total > 0:
            f.associate_device(out["code"].data_ptr(), total, map_codes.data_ptr(), M, a_idx[slot].data_ptr(),
                               a_dist[slot].data_ptr())
        ev_b.record(exts[slot])
        cur.wait_event(ev_b)                        # the map changes only after this batch has been matched
        h3 = time.perf_counter()
        # map update, rank-major / frame-minor so every rank holds the same map
        k = src.shape[1]
        map_codes[args.map: args.map + world * k] = src.reshape(-1, 32)
        ev_c.record(cur)
        exts[slot].wait_event(ev_c)                 # the handle's next batch may overwrite `out` only after it was read
        h4 = time.perf_counter()
        host_ms["wait"] += h1 - h0; host_ms["merge"] += h2 - h1; host_ms["assoc"] += h3 - h2; host_ms["map"] += h4 - h3
        host_ms["n"] += 1

    def run(steps):
        inflight = []
        for k in range(steps):
            slot = k % D
            if len(inflight) == D:
                finish(inflight.pop(0))
            fes[slot].submit_device(frames.data_ptr(), B, ptrs[slot], cap, describe=True)
            inflight.append(slot)
        while inflight:
            finish(inflight.pop(0))

    def sync_all():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup)
    sync_all()
    for f in fes:
        f.reset_timing()
        f.set_profiling(True)
    for k_ in ("wait", "merge", "assoc", "map", "n"):
        host_ms[k_] = 0
    t0 = time.perf_counter()
    run(args.steps)
    sync_all()
    dt = time.perf_counter() - t0
    host_profile = {k_: round(1e3 * v / max(host_ms["n"], 1), 3) for k_, v in host_ms.items() if k_ != "n"}
    for f in fes:
        f.set_profiling(False)
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    def collect():
        t_ = {}
        for f in fes:
            for name, (ms, launches) in f.timing().items():
                a0, b0 = t_.get(name, (0.0, 0))
                t_[name] = (a0 + ms, b0 + launches)
        return t_

    timing_overlapped = collect()       # event durations inside the timed region (batches overlap -> kernels share the GPU)
    # per-kernel durations for the roofline: the same step, one batch in flight, HIP events on the
    # launch stream, right after the timed region (so a kernel's time is not inflated by a neighbour)
    solo_steps = max(3, min(10, args.steps))
    for f in fes:
        f.reset_timing()
    fes[0].set_profiling(True)
    for _ in range(solo_steps):
        fes[0].submit_device(frames.data_ptr(), B, ptrs[0], cap, describe=True)
        finish(0)
    sync_all()
    fes[0].set_profiling(False)
    timing = collect()

    result = None
    if rank == 0:
        frames_total = world * B * args.steps
        value = frames_total / dt
        sb = stage_bytes(P, Ps, cfg["img_size"] != cfg["in_size"])
        kernels = []
        for name, (ms, launches) in timing.items():
            if launches == 0:
                continue
            avg = ms / launches
            e = {"stage": name, "avg_ms": round(avg, 4), "launches": launches}
            if name in sb and avg > 0:
                e["algorithmic_bytes"] = sb[name] * B
                e["GBps"] = round(sb[name] * B / (avg * 1e-3) / 1e9, 1)
            if name == "assoc_mfma" and avg > 0 and seg_total[0] > 0:
                # 2 * N * M * 256 int8 ops per call against the dense int8 MFMA peak (stage time
                # includes the +-64 packing of queries and map and the final decode)
                ops = 2.0 * seg_total[0] * M * 256
                e["algorithmic_ops"] = ops
                e["Pop_per_s"] = round(ops / (avg * 1e-3) / 1e15, 3)
                e["mfma_peak_Pop_per_s"] = INT8_MFMA_PEAK_POPS
                e["frac_of_mfma_peak"] = round(ops / (avg * 1e-3) / 1e15 / INT8_MFMA_PEAK_POPS, 3)
            kernels.append(e)
        streaming = [k for k in kernels if "GBps" in k]
        # dominant streaming kernel = the one that has to move the most bytes
        dom = max(streaming, key=lambda k: k["algorithmic_bytes"]) if streaming else None
        roofline = None
        if dom:
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
            if os.path.exists(tpath):
                tj = json.load(open(tpath))
                if tj["workload"] == {"batch": B, "geometry": args.geometry}:
                    traffic = tj["traffic_bytes_per_launch"].get(dom["stage"])
            roofline = {"bound": "hbm", "kernel": dom["stage"], "achieved": dom["GBps"], "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(dom["GBps"] / HBM_PEAK_GBS, 4), "traffic": traffic,
                        "avg_launch_ms": dom["avg_ms"], "algorithmic_bytes_per_launch": dom["algorithmic_bytes"],
                        "measured": "HIP events on the launch stream, %d single-batch steps run right after the "
                                    "timed region (per-kernel times inside the overlapped region are in "
                                    "kernels_timed_region)" % solo_steps,
                        "traffic_source": "profiles/r01_traffic.json (rocprofv3 PMC, FETCH_SIZE x2 + WRITE_SIZE)" if traffic else None,
                        "note": "dominant STREAMING kernel (most algorithmic bytes).  The longest kernel by time, lsd_grow "
                                "(sequential-semantics LSD region growing, one wave per problem), is latency / issue bound and "
                                "has no bandwidth or MFMA roofline (SURVEY 8d: report time); its time is in `kernels`, its "
                                "instruction profile in DESIGN.md section 9"}
        result = {
            "metric": "frames/sec (%dx%d) detect->descript->project->sanity->associate" % (in_cols, in_rows),
            "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8/f64 (i8 MFMA for association)", "data": "synthetic",
            "config": {"workload": "%s%d-frame batch per GPU of %dx%d synthetic lane frames, %s geometry (working image %dx%d, "
                                   "LSD image %dx%d), LSD+LBD+project+sanity, Hamming association vs %d-code live map"
                                   % ("BASELINE configs[4] frame size (optional stress mode): " if hd else "BASELINE configs[1]: ",
                                      B, in_cols, in_rows, args.geometry, fe.cols, fe.rows, fe.lsd_cols, fe.lsd_rows, M),
                       "frames_per_gpu_per_step": B, "segments_per_step_rank0": seg_total[0],
                       "host_ms_per_step": host_profile, "batches_in_flight": D, "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                       "parallelism": "frame-sharded x%d, all-gather of segment blocks" % world},
            "roofline": roofline,
            "kernels": kernels,
            "kernels_timed_region": [{"stage": n, "avg_ms": round(ms / max(l, 1), 4), "launches": l}
                                     for n, (ms, l) in timing_overlapped.items() if l],
        }
        # ---- CPU baseline: the oracle (single-threaded restatement of the reference path)
        if world == 1 and args.cpu_frames >= 0:
            from oracle.oracle import Oracle
            o = Oracle(cfg)
            nf = args.cpu_frames or (160 if args.geometry == "fullres" else (24 if hd else 1500))
            nf = min(nf, B)
            mc = map_codes.cpu().numpy()
            # parity gate of this very run: the GPU output of the last batch on handle 0,
            # frame by frame, against what the oracle computes for the same frames while it is being timed
            g_fo = outs[0]["frame_offset"].cpu().numpy()
            g_n = int(g_fo[-1])
            g = {k_: outs[0][k_][:g_n].cpu().numpy() for k_ in ("lines", "normals", "color", "pixels_normalized", "ground", "keep", "code")}
            g_desc = outs[0]["desc"][:g_n].cpu().numpy()
            gate_ok, gate_frames, gate_desc_err = True, 0, 0.0
            cdt = 0.0
            nseg = 0
            for f in range(nf):
                t1 = time.perf_counter()
                r = o.process_frame(host[f], cap=3 * args.cap)
                nseg += r["n"]
                if r["n"]:
                    o.match(r["code"], mc)
                cdt += time.perf_counter() - t1            # the comparison below is not part of the baseline
                a_, b_ = int(g_fo[f]), int(g_fo[f + 1])
                same = (b_ - a_) == r["n"] and all(np.array_equal(g[k_][a_:b_], r[k_]) for k_ in g)
                if same and r["n"]:
                    gate_desc_err = max(gate_desc_err, float(np.abs(g_desc[a_:b_] - r["desc"]).max()))
                gate_ok = gate_ok and same
                gate_frames += 1
            result["parity_gate"] = {
                "frames": gate_frames, "identical": bool(gate_ok and gate_desc_err <= 1e-4),
                "fields_bit_exact": ["lines", "normals", "color", "pixels_normalized", "ground", "keep", "code"],
                "descriptor_max_abs_diff": gate_desc_err, "descriptor_tolerance": 1e-4,
                "what": "GPU output of the run's last batch vs the oracle on the same frames"}
            result["cpu_baseline"] = {
                "value": round(nf / cdt, 2), "unit": "frames/s", "cores": 1, "kind": "port",
                "sample": "%d of the %d frames of one step (same synthetic frames, same config, same %d-code map), "
                          "oracle/liblforacle.so single thread, %.1f s" % (nf, B, M, cdt),
                "host_cpus": os.cpu_count()}
            # the same port on many host cores at once (frames are independent; ctypes releases the GIL):
            # BASELINE.md section 3 asks for both the single-thread and the all-cores figure
            from concurrent.futures import ThreadPoolExecutor
            T = max(1, min(64, (os.cpu_count() or 1) // 2))
            per = 8 if args.geometry == "fullres" else (1 if hd else 80)
            oracles = [Oracle(cfg) for _ in range(T)]

            def work(t):
                for k in range(per):
                    r = oracles[t].process_frame(host[(t * per + k) % B], cap=3 * args.cap)
                    if r["n"]:
                        oracles[t].match(r["code"], mc)

            t2 = time.perf_counter()
            with ThreadPoolExecutor(T) as ex:
                list(ex.map(work, range(T)))
            mdt = time.perf_counter() - t2
            result["cpu_baseline_threads"] = {
                "value": round(T * per / mdt, 1), "unit": "frames/s", "cores": T, "kind": "port",
                "sample": "%d threads x %d frames of the same step, one oracle instance per thread, %.1f s" % (T, per, mdt)}
    # The JSON line is the LAST thing on stdout: libraries that log through C stdio (RCCL prints its version banner
    # under NCCL_DEBUG=VERSION, buffered until exit when stdout is a pipe) are flushed on every rank first.
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    if multi:
        dist.barrier()
    for f in fes:
        f.close()
    if multi:
        dist.destroy_process_group()
        ctypes.CDLL(None).fflush(None)
    if rank == 0:
        print(json.dumps(result))
        sys.stdout.flush()
    return result


if __name__ == "__main__":
    main()
