#!/usr/bin/env python3
"""Headline benchmark: frames/s through detect -> describe -> ground-project -> sanity ->
associate on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the whole hot path over one batch of 256 synthetic 640x480 BGR frames
per GPU, already resident in HBM (BASELINE.json configs[1], full-res geometry: img_size
[480,640], top_cutoff 160 -> 640x320 working image): detect -> describe -> project -> sanity, then the
package's associator (lane_slam_amd.LineAssociator / lf_map_*): Hamming association of the step's descriptors
against a device-resident live map of 66 384 entries, and the map update with the kept segments.  With N > 1 GPUs
frames shard across ranks (weak scaling, 256 frames per GPU per step); the per-rank segment blocks are merged with
ONE RCCL all-gather per step (lane_slam_amd.distributed.ShardedAssociator, the class the gloo test covers) and every
rank applies the same blocks to its map replica; association itself needs no collective.

Rank 0 prints ONE JSON line.  `roofline` describes the dominant streaming kernel with live HIP-event timings from
the library's own stream; `cpu_baseline` times the CPU oracle (oracle/, a single-threaded restatement of the
reference path) on a bounded sample of the same workload on the GPU box's host.  `secondary` holds numbers that
never replace `value`: the BASELINE configs[2] stream replay (host-resident frames, per-batch H2D, growing map), the
JPEG-ingest rate, the single-frame plugin latency at the reference's own operating point, the configs[4] associator
stress.
"""
import argparse
import json
import os
import sys
import time

# Eight batches (six until round 4) are kept in flight on as many HIP streams (plus the map's and torch's); ROCclr maps streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share a queue serialise.  Must be
# set before the HIP runtime initialises.  32: the content rows keep up to 18 batches in flight (lf_suggested_depth) and a handle
# that shares its queue with another stream loses its overlap (16 batches on 16 queues measured 15 % slower than on 24).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
# kernel arguments in device memory instead of host-coherent memory: every kernel starts ~2 us sooner (its first scalar
# loads no longer cross PCIe); matters for the latency numbers, not for the pipelined rate
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
# dmabuf IPC (the only kind the pool's host driver supports) -- read when HSA initialises, so it is set here, before
# any torch GPU call, not next to init_process_group
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

INT8_MFMA_PEAK_POPS = 5.0     # dense int8 MFMA, MI355X_MICROARCH.md (2x the ~2.5 PF bf16 rate)
FP4_MFMA_PEAK_POPS = 10.0     # dense FP4 (f8f6f4 with e2m1 operands), MI355X_MICROARCH.md: 2x the int8 / fp8 rate
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)
MAP_ROLL = 16384               # live map = --map seeded codes + this many more entries, kept full (ring)


def stage_bytes(P):
    """ALGORITHMIC HBM bytes per frame of each streaming kernel, SURVEY.md section 8(d) (P = working pixels).  The LSD
    stages are not listed: their traffic is proportional to the number of edge pixels (sparse records), they are
    latency / issue bound and are reported as time only."""
    return {
        "pre(resize+correct+hsv+masks+dilate)": 3 * P + 3 * P + 3 * P,      # K_pre: read 3P, write 3P + 3P
        "canny_nms": 3 * P + P,                                              # K_canny_grad: read 3P, write P
        "lbd_gray_blur_sobel": P + 4 * P,                                    # K_sobel_lbd: read P (gray), write 4P
    }


def source_digest(*rel):
    """sha256 (16 hex digits) of library sources: ties a PMC traffic file to the kernel it was measured on"""
    import hashlib
    h = hashlib.sha256()
    for r in rel:
        with open(os.path.join(ROOT, r), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def alloc_out(torch, dev, B, cap):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--batch", type=int, default=0, help="frames per GPU per step (0: 256, or 128 for --geometry hd)")
    ap.add_argument("--geometry", default="fullres", choices=["fullres", "parity", "hd"],
                    help="fullres / parity: BASELINE configs[1] (640x480 input); hd: configs[4]'s 1920x1080 frames "
                         "(img_size [1080,1920], top_cutoff 360) -- an optional stress mode, not the headline")
    ap.add_argument("--map", type=int, default=50000, help="seeded live-map codes (the map holds these + 16384 more entries)")
    ap.add_argument("--unique", type=int, default=256, help="distinct synthetic frames per rank (tiled to --batch)")
    ap.add_argument("--cap", type=int, default=512, help="max lines per (frame, colour)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo (host staging) only to dry-run the N>1 path on a shared GPU")
    ap.add_argument("--depth", type=int, default=0, help="independent batches in flight (handles/streams); 0: 8 (6 until round 4)")
    ap.add_argument("--seed-order", default="opencv32", choices=["opencv30", "opencv32"],
                    help="LSD seed order inside a gradient bin (lf_config.lsd_seed_order): opencv32 = std::sort's, as on ROS Kinetic's 3.3.1 "
                         "(what the reference computes: the default); opencv30 = raster order, an A/B option")
    ap.add_argument("--tie-rule", default="mihasher", choices=["mihasher", "lowest"],
                    help="which of several equally near map entries the live map returns: mihasher = the one "
                         "BinaryDescriptorMatcher::match finds first (the default), lowest = the lowest index (one pass, an A/B option)")
    ap.add_argument("--lsd-refine", type=int, default=2, help="diagnostic only: 0/1 skip refine / NFA stages (invalid as a headline run)")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames for the CPU baseline (0 = auto, -1 = skip)")
    ap.add_argument("--secondary", default="auto", choices=["auto", "all", "none"],
                    help="secondary measurements (stream replay, JPEG ingest, plugin latency, associator stress): auto = all at "
                         "N = 1 with the default geometry, none otherwise")
    ap.add_argument("--stream-frames", type=int, default=1024, help="frames of the configs[2] stream (replayed --stream-laps times)")
    ap.add_argument("--stream-laps", type=int, default=12)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from lane_slam_amd import FrontEnd, LineAssociator, default_config, synth, _lib
    from lane_slam_amd.distributed import ShardedAssociator

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
    # LF_FORCE_COLLECTIVES=1 runs the N > 1 code path (process group, all-gather, block merge) with a single
    # rank: a dry run of the RCCL calls on one GPU, never a headline configuration
    force = bool(os.environ.get("LF_FORCE_COLLECTIVES"))
    multi = world > 1 or force
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
    D = args.depth if args.depth > 0 else 8
    in_rows, in_cols = (1080, 1920) if hd else (480, 640)
    cfg = default_config("fullres", in_size=(in_rows, in_cols)) if hd else default_config(args.geometry)
    cfg["lsd"]["refine"] = args.lsd_refine
    cfg["lsd"]["seed_order"] = args.seed_order
    # D handles = D independent batches in flight (each handle owns a HIP stream and its buffers: the per-problem lists of the LSD stages
    # start at an eighth of the LSD image and grow with the content -- 2.4 GB at 640x480 x 256 frames on lane frames);
    # with --depth 0 no more of them than fit 85 % of the device's free memory, measured on the first handle
    free0, _total_b = torch.cuda.mem_get_info(local_rank)
    fes = [FrontEnd(cfg, device=local_rank, max_frames=B, max_lines_per_color=args.cap)]
    handle_bytes = max(1, free0 - torch.cuda.mem_get_info(local_rank)[0])
    # ... and budgeted as GROWN: the lists of a handle that meets busier content are reallocated larger by lf_wait (camera frames: 2.75 x;
    # 118 bytes per list entry and problem (102 until the seed sums of round 6)) -- room for a fourfold growth of every handle, so that a growth in the
    # middle of a run does not meet a device filled to the brim by handles sized before any growth
    S0 = fes[0].lsd_list_capacity()[0]
    grown_bytes = handle_bytes + 118 * 3 * B * max(0, min(fes[0].lsd_rows * fes[0].lsd_cols, 4 * S0) - S0)
    if args.depth == 0:
        D = max(2, min(D, int((0.85 * free0) // grown_bytes)))
    args.handle_bytes, args.grown_bytes = handle_bytes, grown_bytes          # (secondary(): the content rows add handles)
    fes += [FrontEnd(cfg, device=local_rank, max_frames=B, max_lines_per_color=args.cap) for _ in range(D - 1)]
    fe = fes[0]
    P = fe.rows * fe.cols

    # ---- synthetic input, resident in HBM before the timed region
    uniq = min(args.unique, B)
    host_threads = max(1, min(32, (os.cpu_count() or 2) // 2))
    host = synth.make_batch(uniq, seed0=10000 * rank, rows=in_rows, cols=in_cols, threads=host_threads)
    reps = (B + uniq - 1) // uniq
    host = np.ascontiguousarray(np.tile(host, (reps, 1, 1, 1))[:B])
    # one device-resident input batch PER HANDLE (D distinct buffers, the frames of batch k shifted sideways by 9 k pixels):
    # consecutive steps read different memory, so no step finds its input in the 256 MiB Infinity Cache because the
    # previous one left it there (one shared 236 MB tensor could not rule that out)
    frames0 = torch.from_numpy(host).to(dev)
    frames_d = [frames0] + [torch.roll(frames0, 9 * k, dims=2).contiguous() for k in range(1, D)]
    frames = frames0

    cap = B * 3 * args.cap
    outs = [alloc_out(torch, dev, B, cap) for _ in range(D)]
    ptrs = [{k: v.data_ptr() for k, v in o.items()} for o in outs]
    a_idx = [torch.zeros(cap, dtype=torch.int32, device=dev) for _ in range(D)]
    a_dist = [torch.zeros(cap, dtype=torch.float32, device=dev) for _ in range(D)]

    # ---- the live map: the package's associator component.  Seeded FULL (--map + 16384 random codes, seed 1234,
    # identical on every rank) and kept full by the ring, so the association work per GPU is the same at every N
    # and in every step (weak scaling); the kept segments of every rank's batch enter it each step
    # (append-only like show_map.py:28-42, oldest entries overwritten).
    M = args.map + MAP_ROLL
    amap = LineAssociator(capacity=M, color_gating=False, max_distance=128, policy="append", kept_only=True,
                          when_full="ring", device=local_rank, tie_rule=args.tie_rule)
    amap.seed(synth.random_codes(M, 1234))
    G = 16 * 1024                                    # segments per rank block; a batch with more raises (never truncates)
    sharded = ShardedAssociator(amap, block_segments=G, device=dev, backend=args.backend, force_collective=force)
    seg_total = [0]
    host_ms = {"wait": 0.0, "associate+exchange+update": 0.0, "n": 0}
    step_no = [0]
    # per-frame odometry poses (map -> duck, odometry.py:110-120): a gentle arc, the same for every batch
    poses = np.column_stack([0.01 * np.arange(B), 0.3 * np.sin(0.02 * np.arange(B)), 0.002 * np.arange(B)])

    def finish(slot):
        """Complete the batch queued on `slot`: associate against the map replica, exchange segment blocks (N > 1),
        update the map.  The only host synchronisations are lf_wait (the batch's segment count) and the map's
        size mirror of the PREVIOUS update; everything else is queued on the map's stream and ordered against the
        handle's stream with events inside the library."""
        f = fes[slot]
        h0 = time.perf_counter()
        total = f.wait()
        h1 = time.perf_counter()
        seg_total[0] = total
        sharded.step(f, outs[slot], total, B, a_idx[slot], a_dist[slot], poses=poses, step=step_no[0])
        step_no[0] += 1
        h2 = time.perf_counter()
        host_ms["wait"] += h1 - h0
        host_ms["associate+exchange+update"] += h2 - h1
        host_ms["n"] += 1

    def run(steps):
        inflight = []
        for k in range(steps):
            slot = k % D
            if len(inflight) == D:
                finish(inflight.pop(0))
            fes[slot].submit_device(frames_d[slot].data_ptr(), B, ptrs[slot], cap, describe=True)
            inflight.append(slot)
        while inflight:
            finish(inflight.pop(0))

    def sync_all():
        amap.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up, before any warm-up step: every handle runs one batch once (buffers grown on demand, kernel attributes, the handle's
    # first look at the workload for its LDS-slice policy) -- with fewer warm-up steps than handles one of them would otherwise meet
    # its first batch inside the timed region
    for slot in range(D):
        fes[slot].submit_device(frames_d[slot].data_ptr(), B, ptrs[slot], cap, describe=True)
    for slot in range(D):
        fes[slot].wait()
    sync_all()
    run(args.warmup)
    sync_all()
    for f in fes:
        f.reset_timing()
        f.set_profiling(True)
    amap.timing()
    amap.set_profiling(True)
    for k_ in host_ms:
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
        for name, (ms, launches) in amap.timing().items():
            t_[name] = (ms, launches)
        return t_

    timing_overlapped = collect()       # event durations inside the timed region (batches overlap -> kernels share the GPU)
    # per-kernel durations for the roofline: the same step, one batch in flight, HIP events on the
    # launch stream, right after the timed region (so a kernel's time is not inflated by a neighbour)
    solo_steps = max(3, min(10, args.steps))
    for f in fes:
        f.reset_timing()
    fes[0].set_profiling(True)
    for _ in range(solo_steps):
        fes[0].submit_device(frames_d[0].data_ptr(), B, ptrs[0], cap, describe=True)
        finish(0)
        amap.synchronize()
    sync_all()
    fes[0].set_profiling(False)
    amap.set_profiling(False)
    timing = collect()
    map_state = amap.state()

    result = None
    if rank == 0:
        frames_total = world * B * args.steps
        value = frames_total / dt
        sb = stage_bytes(P)
        kernels = []
        for name, (ms, launches) in timing.items():
            if launches == 0:
                continue
            avg = ms / launches
            e = {"stage": name, "avg_ms": round(avg, 4), "launches": launches}
            if name in sb and avg > 0:
                e["algorithmic_bytes"] = sb[name] * B
                e["GBps"] = round(sb[name] * B / (avg * 1e-3) / 1e9, 1)
                e["frac_of_hbm_peak"] = round(e["GBps"] / HBM_PEAK_GBS, 4)
            if name == "lsd_blur_resample_grad" and avg > 0:
                # the bytes of the design that was built -- bit planes in (the Canny edge plane and the three colour masks of
                # every frame: P / 8 bytes each), sparse records out (32 bytes per pixel with a defined gradient, 12 per pixel with a
                # non-zero gradient below the threshold when lsd_seed_order = opencv32: counted on the last batch) -- not SURVEY 8d's dense
                # planes, which this stage never writes.  It is bound by its double-precision blur and resampling, not by these bytes.
                try:
                    nd = int(fes[0].fetch(_lib.LF_BUF_LSD_NORDER, B).sum())
                    nl = int(fes[0].fetch(_lib.LF_BUF_LSD_NLOW, B).sum())
                    e["algorithmic_bytes"] = B * 4 * (P // 8) + 32 * nd + 12 * nl
                    e["GBps"] = round(e["algorithmic_bytes"] / (avg * 1e-3) / 1e9, 1)
                    e["frac_of_hbm_peak"] = round(e["GBps"] / HBM_PEAK_GBS, 4)
                    e["bytes_note"] = "bit planes in (4 P / 8 per frame) + %d records x 32 B + %d low-gradient records x 12 B of one batch; k_lsd_classify + k_lsd_grad" % (nd, nl)
                except Exception as ex:
                    e["bytes_note"] = "record counts unavailable: %r" % (ex,)
            if name == "assoc_mfma" and avg > 0 and seg_total[0] > 0:
                # 2 * N * M * 256 ops per call (one launch: query expansion, matrix loop, merge, report; the
                # map's operands stay packed across calls), priced against the dense peak OF THE INSTRUCTION IT RUNS ON
                ops = 2.0 * seg_total[0] * M * 256
                fp4 = os.environ.get("LF_ASSOC_INT8") is None
                peak = FP4_MFMA_PEAK_POPS if fp4 else INT8_MFMA_PEAK_POPS
                e["algorithmic_ops"] = ops
                e["Pop_per_s"] = round(ops / (avg * 1e-3) / 1e15, 3)
                e["instruction"] = "v_mfma_scale_f32_32x32x64_f8f6f4 (e2m1 operands)" if fp4 else "v_mfma_i32_32x32x32_i8"
                e["mfma_peak_Pop_per_s"] = peak
                e["frac_of_mfma_peak"] = round(ops / (avg * 1e-3) / 1e15 / peak, 3)
            counted = {"pre(resize+correct+hsv+masks+dilate)": ("k_pre", "lane_slam_amd/csrc/k_pre.hip"), "canny_nms": ("k_canny_nms", "lane_slam_amd/csrc/k_canny.hip"),
                       "lbd_gray_blur_sobel": ("k_lbd_grad", "lane_slam_amd/csrc/k_lbd.hip")}
            if name in counted:
                # what bounds a streaming kernel that sits below the HBM roofline: its vector issue utilisation (rocprofv3 SQ passes,
                # tools/profile_round.sh), quoted only while the kernel source is the one the counters were measured on
                kpath = os.path.join(ROOT, "profiles", "kernel_counters.json")
                if os.path.exists(kpath):
                    kj = json.load(open(kpath)).get("kernels", {}).get(counted[name][0])
                    if kj and kj.get("source_digest") == source_digest(counted[name][1]):
                        e["counters"] = {k_: kj[k_] for k_ in ("valu_issue_utilisation", "avg_resident_waves", "frac_wave_cycles_issue_stalled") if k_ in kj}
            if name == "lsd_grow":
                # latency / issue bound: no bandwidth or matrix roofline.  What bounds it, as counters: wave-slot
                # occupancy, vector issue utilisation and where its wave-cycles go (rocprofv3 SQ passes measured offline by
                # tools/profile_round.sh; quoted only while the kernel source is the one they were measured on)
                gpath = os.path.join(ROOT, "profiles", "grow_counters.json")
                if os.path.exists(gpath):
                    gj = json.load(open(gpath))
                    if gj.get("source_digest") == source_digest("lane_slam_amd/csrc/lsd_grow.h", "lane_slam_amd/csrc/k_lsd_grow.hip"):
                        e["counters"] = {k_: v for k_, v in gj.items() if k_ in ("one_batch_in_flight", "six_batches_in_flight", "source")}
                    else:
                        e["counters"] = "profiles/grow_counters.json was measured on another lsd_grow.h / k_lsd_grow.hip: not quoted"
            kernels.append(e)
        streaming = [k for k in kernels if "GBps" in k and k["stage"] in sb]
        # dominant streaming kernel = the one that has to move the most bytes
        dom = max(streaming, key=lambda k: k["algorithmic_bytes"]) if streaming else None
        roofline = None
        if dom:
            # PMC traffic is measured offline (separate rocprofv3 --pmc passes, tools/profile_round.sh) and is only quoted
            # while the kernel it was measured on is the kernel that runs: the file records the digest of its source
            traffic, traffic_note = None, None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                tj = json.load(open(tpath))
                if tj["workload"] == {"batch": B, "geometry": args.geometry}:
                    if tj.get("source_digest", {}).get("k_pre") == source_digest("lane_slam_amd/csrc/k_pre.hip"):
                        traffic = tj["traffic_bytes_per_launch"].get(dom["stage"])
                    else:
                        traffic_note = "profiles/traffic.json was measured on another k_pre.hip: not quoted"
            roofline = {"bound": "hbm", "kernel": dom["stage"], "achieved": dom["GBps"], "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(dom["GBps"] / HBM_PEAK_GBS, 4), "traffic": traffic,
                        "avg_launch_ms": dom["avg_ms"], "algorithmic_bytes_per_launch": dom["algorithmic_bytes"],
                        "measured": "HIP events on the launch stream, %d single-batch steps run right after the "
                                    "timed region (per-kernel times inside the overlapped region are in "
                                    "kernels_timed_region)" % solo_steps,
                        "traffic_source": "profiles/traffic.json (rocprofv3 PMC passes, corrected as MI355X_MICROARCH.md prescribes)" if traffic else traffic_note,
                        "note": "dominant STREAMING kernel (most algorithmic bytes, SURVEY 8d).  The longest kernel by time, lsd_grow "
                                "(sequential-semantics LSD region growing), is latency / issue bound and "
                                "has no bandwidth or MFMA roofline (SURVEY 8d: report time); its time is in `kernels`"}
        result = {
            "metric": "frames/sec (%dx%d) detect->descript->project->sanity->associate" % (in_cols, in_rows),
            "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8/f64 (FP4 e2m1 MFMA with exact f32 accumulation for association)", "data": "synthetic",
            "config": {"workload": "%s%d-frame batch per GPU of %dx%d synthetic lane frames (%d distinct), %s geometry (working image %dx%d, "
                                   "LSD image %dx%d), LSD+LBD+project+sanity, Hamming association vs the %d-entry live map + map update; %d distinct device-resident input batches in rotation"
                                   % ("BASELINE configs[4] frame size (optional stress mode): " if hd else "BASELINE configs[1]: ",
                                      B, in_cols, in_rows, uniq, args.geometry, fe.cols, fe.rows, fe.lsd_cols, fe.lsd_rows, M, D),
                       "frames_per_gpu_per_step": B, "segments_per_step_rank0": seg_total[0], "live_map": map_state,
                       "lsd_seed_order": args.seed_order, "tie_rule": args.tie_rule,
                       "handle_gb": round(handle_bytes / 1e9, 2), "lsd_list_entries": fe.lsd_list_capacity()[0], "lsd_lists_grown": sum(f.lsd_list_capacity()[1] for f in fes),
                       "host_ms_per_step": host_profile, "batches_in_flight": D, "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                       "parallelism": "frame-sharded x%d, one all-gather of segment blocks per step, replicated map" % world},
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
            mc = amap.fetch(0, M)["code"]
            # parity gate of this very run: the GPU output of the last batch on handle 0,
            # frame by frame, against what the oracle computes for the same frames while it is being timed
            g_fo = outs[0]["frame_offset"].cpu().numpy()
            g_n = int(g_fo[-1])
            g = {k_: outs[0][k_][:g_n].cpu().numpy() for k_ in ("lines", "normals", "color", "pixels_normalized", "ground", "keep", "code")}
            g_desc = outs[0]["desc"][:g_n].cpu().numpy()
            gate_ok, gate_frames, gate_desc_err = True, 0, 0.0
            cdt = 0.0
            for f in range(nf):
                t1 = time.perf_counter()
                r = o.process_frame(host[f], cap=3 * args.cap)
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
        do_secondary = args.secondary == "all" or (args.secondary == "auto" and world == 1 and not force and args.geometry == "fullres")
        if do_secondary:
            result["secondary"] = secondary(args, torch, dev, local_rank, fes, ptrs, a_idx, a_dist, host, B, D, cap)
            if args.seed_order == "opencv32" and args.tie_rule == "mihasher":
                # the other configuration (OpenCV 3.0 / 3.1 seed order, lowest-index ties: NOT what the reference's stack computes),
                # side by side: the same bench as a child process after this one's buffers are gone (below)
                result["secondary"]["ab_opencv30_lowest"] = "pending"
    # The JSON line is the LAST thing on stdout: libraries that log through C stdio (RCCL prints its version banner
    # under NCCL_DEBUG=VERSION, buffered until exit when stdout is a pipe) are flushed on every rank first.
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    if multi:
        dist.barrier()
    amap.close()
    for f in fes:
        f.close()
    if multi:
        dist.destroy_process_group()
        ctypes.CDLL(None).fflush(None)
    if rank == 0 and isinstance(result.get("secondary"), dict) and result["secondary"].get("ab_opencv30_lowest") == "pending":
        import subprocess
        try:
            torch.cuda.empty_cache()
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--steps", str(max(20, args.steps)), "--warmup", str(args.warmup),
                                  "--seed-order", "opencv30", "--tie-rule", "lowest", "--secondary", "none", "--cpu-frames", "-1",
                                  "--batch", str(args.batch), "--map", str(args.map)], capture_output=True, text=True, timeout=600)
            ab = json.loads(out.stdout.strip().splitlines()[-1])
            result["secondary"]["ab_opencv30_lowest"] = {
                "value": ab["value"], "unit": "frames/s", "ms_per_step": ab["ms_per_step"], "steps": ab["steps"],
                "note": "lsd_seed_order = opencv30 (raster order inside a bin: OpenCV 3.0 / 3.1) and tie_rule = lowest (lowest map index): the "
                        "A/B options, NOT the reference's behaviour (ROS Kinetic's OpenCV 3.3.1 orders seeds with std::sort; "
                        "BinaryDescriptorMatcher::match returns the first-discovered candidate); same bench, child process, same GPU"}
        except Exception as e:                                           # a measurement beside the headline: never fatal
            result["secondary"]["ab_opencv30_lowest"] = {"error": str(e)[:200]}
    if rank == 0:
        print(json.dumps(result))
        sys.stdout.flush()
    return result


def secondary(args, torch, dev, device_id, fes, ptrs, a_idx, a_dist, host, B, D, cap):
    """Measurements that never replace `value`.  Each is bounded to a few seconds."""
    from lane_slam_amd import FrontEnd, LineAssociator, LineDetectorHIP, synth
    from lane_slam_amd.config import DEFAULT_DETECTOR_CONFIGURATION
    sec = {}

    # ---- BASELINE configs[2]: streaming replay.  Frames live in PINNED HOST memory and every batch pays its own
    # H2D copy on its handle's stream (D batches in flight = multi-buffered H2D); distinct frames; the map starts
    # EMPTY and grows by the kept segments (append-only, show_map.py:28-42), with per-frame poses.
    n_stream = max(B, (args.stream_frames // B) * B)
    laps = max(1, args.stream_laps)
    uniq = host.shape[0]
    pinned = torch.empty((n_stream,) + host.shape[1:], dtype=torch.uint8).pin_memory()
    pn = pinned.numpy()
    for j in range(n_stream):
        # the `uniq` rendered frames, shifted sideways by 9 px per pass: cheap, and every frame of the stream differs
        pn[j] = np.roll(host[j % uniq], 9 * (j // uniq), axis=1)
    frame_bytes = host[0].nbytes
    segs = [0]

    Ds = min(D, 6)        # host-fed batches: the copies share one PCIe link, six in flight measured better than eight (84 k / 79 k frames/s)

    def stream_pass(smap, lap0, n_laps):
        """n_laps replays of the stream, back to back: the pipeline is filled once and drained once"""
        inflight = []

        def complete():
            s0, k0, lap = inflight.pop(0)
            n = fes[s0].wait()
            segs[0] += n
            t = np.arange(k0 * B, (k0 + 1) * B, dtype=np.float64) + lap * n_stream
            pose = np.column_stack([0.01 * t, 0.3 * np.sin(0.02 * t), 0.002 * t])
            smap.step_device(fes[s0], ptrs[s0], n, B, a_idx[s0].data_ptr(), a_dist[s0].data_ptr(), poses=pose, step=lap * 1000 + k0)
        j = 0
        for lap in range(lap0, lap0 + n_laps):
            for k in range(n_stream // B):
                slot = j % Ds
                j += 1
                if len(inflight) == Ds:
                    complete()
                fes[slot].submit_host(pinned.data_ptr() + k * B * frame_bytes, B, ptrs[slot], cap, describe=True)
                inflight.append((slot, k, lap))
        while inflight:
            complete()

    def new_map():
        return LineAssociator(capacity=1 << 19, color_gating=False, max_distance=128, policy="append", kept_only=True,
                              when_full="ring", device=device_id)
    smap = new_map()
    stream_pass(smap, 0, 1)                          # warm-up lap (allocations, first touch of the pinned pages)
    smap.synchronize()
    torch.cuda.synchronize()
    smap.close()
    smap = new_map()
    segs[0] = 0
    t0 = time.perf_counter()
    stream_pass(smap, 0, laps)
    smap.synchronize()
    torch.cuda.synchronize()
    sdt = time.perf_counter() - t0
    st = smap.state()
    smap.close()
    # lf_process_batch copies the source rows from top_cutoff down (no resize in these geometries: img_size == frame size)
    scfg = fes[0].cfg
    copied_bytes = frame_bytes - int(scfg["top_cutoff"]) * host.shape[2] * 3 if list(scfg["img_size"]) == list(host.shape[1:3]) else frame_bytes
    sec["stream_configs2"] = {
        "value": round(laps * n_stream / sdt, 1), "unit": "frames/s",
        "what": "BASELINE configs[2]: %d-frame stream x %d laps, frames in pinned host memory, one async H2D copy per %d-frame batch "
                "on its handle's stream (%d batches in flight; only the source rows the working image reads are copied: rows "
                ">= top_cutoff, %d of %d bytes per frame), detect->describe->project->sanity->associate->map update, map "
                "growing from empty by the kept segments (append, per-frame poses)" % (n_stream, laps, B, Ds, copied_bytes, frame_bytes),
        "frames": laps * n_stream, "distinct_frames_per_lap": n_stream, "seconds": round(sdt, 4),
        "h2d_GBps": round(laps * n_stream * copied_bytes / sdt / 1e9, 2), "segments": segs[0], "map_final": st}

    # ---- other content: the rate of region growing depends on what is in the frames.  (a) the three real Duckiebot
    # camera frames committed as test inputs (tests/golden/real_frames.npz), tiled to a batch, each copy shifted sideways;
    # (b) the synthetic frames with clutter (speckle + 40 random strokes in lane colours per frame: many short regions)
    extra = {"fes": [], "ptrs": []}

    def content_rate(batch_host, label):
        d = torch.from_numpy(np.ascontiguousarray(batch_host)).to(dev)
        torch.cuda.synchronize()
        state = {"D": D}

        def go(nb):
            Dn = state["D"]
            hs, ps = fes + extra["fes"], ptrs + extra["ptrs"]
            inflight, segsum = [], 0
            for k in range(nb):
                slot = k % Dn
                if len(inflight) == Dn:
                    segsum += hs[inflight.pop(0)].wait()
                hs[slot].submit_device(d.data_ptr(), B, ps[slot], cap, describe=True)
                inflight.append(slot)
            while inflight:
                segsum += hs[inflight.pop(0)].wait()
            return segsum
        go(D)
        torch.cuda.synchronize()
        # the in-flight depth follows the workload (lf_suggested_depth): busy content is bound by the chain of its longest
        # problems, more batches in flight fill the machine; the extra handles are created once and kept for the other rows
        want = max(D, fes[0].suggested_depth()) if args.depth == 0 else D
        while len(fes) + len(extra["fes"]) < want:
            if torch.cuda.mem_get_info(device_id)[0] < 2 * args.grown_bytes + (len(fes) + len(extra["fes"])) * (args.grown_bytes - args.handle_bytes):
                want = len(fes) + len(extra["fes"])          # no room for one more handle AND the growth of those that exist
                break
            extra["fes"].append(FrontEnd(fes[0].cfg, device=device_id, max_frames=B, max_lines_per_color=args.cap))
            o_ = alloc_out(torch, dev, B, cap)
            extra.setdefault("outs", []).append(o_)
            extra["ptrs"].append({k: v.data_ptr() for k, v in o_.items()})
        state["D"] = want
        if want > D:
            go(want)
            torch.cuda.synchronize()
        D_used = want
        nb = 8 * D_used                                    # long enough for the fill and the drain of the pipeline (one batch's latency each) to be a few per cent
        t0 = time.perf_counter()
        n_seg = go(nb)
        torch.cuda.synchronize()
        cdt = time.perf_counter() - t0
        return {"value": round(nb * B / cdt, 1), "unit": "frames/s", "segments_per_frame": round(n_seg / (nb * B), 1),
                "batches_in_flight": D_used,
                "what": "%s; detect->describe->project->sanity (no association), %d batches of %d frames, %d in flight (lf_suggested_depth)" % (label, nb, B, D_used)}
    try:
        # 28 of the reference's camera frames, from their JPEG streams (tests/golden/real_jpegs.npz), decoded by the device decoder
        zj = np.load(os.path.join(ROOT, "tests", "golden", "real_jpegs.npz"))
        streams = [bytes(zj["jpeg%02d" % k]) for k in range(len(zj["names"]))]
        rf, st_ = fes[0].decode_jpeg_batch(streams, n_threads=4)
        rf = [rf[k] for k in range(len(streams)) if st_[k] == 0 and rf[k].shape == host.shape[1:]]
        if rf:
            tiled = np.stack([np.roll(rf[i % len(rf)], 7 * (i // len(rf)), axis=1) for i in range(B)])
            sec["real_frames"] = content_rate(tiled, "%d real Duckiebot camera frames of five recording sessions (the reference's anti_instagram annotation images, tests/golden/real_jpegs.npz) tiled to %d, each copy shifted by 7 px" % (len(rf), B))
        real = np.load(os.path.join(ROOT, "tests", "golden", "real_frames.npz"))
        rf = [real[k] for k in real.files if real[k].ndim == 3 and real[k].shape == host.shape[1:]]
        if rf:
            tiled = np.stack([np.roll(rf[i % len(rf)], 7 * (i // len(rf)), axis=1) for i in range(B)])
            sec["real_frames_r03_sample"] = content_rate(tiled, "the THREE camera frames rounds 2 and 3 measured (tests/golden/real_frames.npz) tiled to %d, each copy shifted by 7 px" % B)
    except Exception as e:
        sec["real_frames"] = {"error": repr(e)}
    try:
        rng_c = np.random.default_rng(4321)
        cl = host.copy()
        for f_ in range(cl.shape[0]):
            img = cl[f_]
            r0 = img.shape[0] // 3
            for _ in range(40):
                y, x = rng_c.integers(r0 + 10, img.shape[0] - 10), rng_c.integers(10, img.shape[1] - 10)
                dy, dx = rng_c.integers(-12, 13), rng_c.integers(-40, 41)
                col = ((235, 235, 235), (40, 220, 235), (40, 40, 220))[rng_c.integers(0, 3)]
                t_ = np.linspace(0, 1, 80)
                yy = np.clip((y + t_ * dy + rng_c.normal(0, 0.7, 80)).astype(int), r0, img.shape[0] - 1)
                xx = np.clip((x + t_ * dx + rng_c.normal(0, 0.7, 80)).astype(int), 0, img.shape[1] - 1)
                img[yy, xx] = col
            img[rng_c.random(img.shape[:2]) < 0.004] = (235, 235, 235)
        sec["clutter_frames"] = content_rate(cl, "the step's synthetic frames + speckle and 40 random strokes in lane colours each (tools/soak_parity.py's clutter)")
    except Exception as e:
        sec["clutter_frames"] = {"error": repr(e)}

    # ---- the second detector: EDLines over 3 octaves + LBD on the detector's gradients, same frames
    try:
        import ctypes as ct
        from lane_slam_amd import _lib as L
        kcap = B * 512
        kout = {k: torch.zeros((kcap, c) if c > 1 else kcap, dtype={"f4": torch.float32, "i4": torch.int32, "u1": torch.uint8}[dt], device=dev)
                for k, dt, c in L.KEYLINE_FIELDS}
        kfo = torch.zeros(B + 1, dtype=torch.int32, device=dev)
        ks = L.LfKeylines()
        ks.capacity = kcap
        ks.frame_offset = kfo.data_ptr()
        for k, _, _ in L.KEYLINE_FIELDS:
            setattr(ks, k, kout[k].data_ptr())
        d = torch.from_numpy(np.ascontiguousarray(host)).to(dev)
        ktot = ct.c_int()

        def kl():
            fes[0]._check(fes[0].lib.lf_keylines_batch(fes[0].h, ct.c_void_p(d.data_ptr()), B, 0, 1, 3, None, ct.byref(ks), 1, 1, ct.byref(ktot), None))
        kl(); kl()
        t0 = time.perf_counter()
        for _ in range(5):
            kl()
        kdt = (time.perf_counter() - t0) / 5
        sec["edlines_keylines"] = {"value": round(B / kdt, 1), "unit": "frames/s", "ms_per_batch": round(kdt * 1e3, 3), "keylines_per_frame": round(ktot.value / B, 1),
                                   "what": "lf_keylines_batch: EDLines over 3 octaves + multi-octave LBD (BinaryDescriptor::operator()), %d frames resident, ONE batch in "
                                           "flight (synchronous call)" % B}
        # the same, pipelined: lf_keylines_batch_async on D handles in flight (one output block per handle)
        kouts = [kout] + [{k: torch.zeros_like(v) for k, v in kout.items()} for _ in range(D - 1)]
        kfos = [kfo] + [torch.zeros_like(kfo) for _ in range(D - 1)]
        kptrs = [dict({k: v.data_ptr() for k, v in o_.items()}, frame_offset=f_.data_ptr()) for o_, f_ in zip(kouts, kfos)]

        def kl_go(nb):
            inflight, tot = [], 0
            for k_ in range(nb):
                slot = k_ % D
                if len(inflight) == D:
                    tot += fes[inflight.pop(0)].wait()
                fes[slot].keylines_submit_device(d.data_ptr(), B, kptrs[slot], kcap, n_octaves=3)
                inflight.append(slot)
            while inflight:
                tot += fes[inflight.pop(0)].wait()
            return tot
        kl_go(D)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ktot2 = kl_go(8 * D)                                      # (4 * D until round 6: as for the ingest row, the fill and drain of the pipeline weighed 3 %)
        torch.cuda.synchronize()
        kdt2 = time.perf_counter() - t0
        sec["edlines_keylines_pipelined"] = {"value": round(8 * D * B / kdt2, 1), "unit": "frames/s", "keylines_per_frame": round(ktot2 / (8 * D * B), 1), "batches_in_flight": D,
                                             "what": "lf_keylines_batch_async + lf_wait: EDLines over 3 octaves + multi-octave LBD, %d batches of %d frames, %d in flight" % (8 * D, B, D)}
    except Exception as e:
        sec["edlines_keylines"] = {"error": repr(e)}

    # ---- EDLines as the detector of the batched end-to-end path (lf_set_detector): detect (EDLines + colour masks) -> normals ->
    # project -> sanity -> LBD, D batches in flight; association left out like in the content rows
    try:
        for f_ in fes:
            f_.set_detector("edlines")
        d = torch.from_numpy(np.ascontiguousarray(host)).to(dev)

        def ed_go(nb):
            inflight, segsum = [], 0
            for k_ in range(nb):
                slot = k_ % D
                if len(inflight) == D:
                    segsum += fes[inflight.pop(0)].wait()
                fes[slot].submit_device(d.data_ptr(), B, ptrs[slot], cap, describe=True)
                inflight.append(slot)
            while inflight:
                segsum += fes[inflight.pop(0)].wait()
            return segsum
        ed_go(D)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nseg = ed_go(6 * D)
        torch.cuda.synchronize()
        edt = time.perf_counter() - t0
        sec["edlines_pipeline"] = {"value": round(6 * D * B / edt, 1), "unit": "frames/s", "segments_per_frame": round(nseg / (6 * D * B), 1), "batches_in_flight": D,
                                   "what": "lf_process_batch_async with LF_DETECTOR_EDLINES: EDLines (one octave) on the gray working image + colour masks -> normals -> "
                                           "project -> sanity -> LBD, the step's synthetic frames, %d batches of %d frames, %d in flight (no association)" % (6 * D, B, D)}
    except Exception as e:
        sec["edlines_pipeline"] = {"error": repr(e)}
    finally:
        for f_ in fes:
            f_.set_detector("lsd")

    # ---- JPEG ingest (8f-1): CompressedImage streams -> host Huffman decode -> GPU IDCT/colour -> the same path
    try:
        import io
        from PIL import Image
        streams = []
        for i in range(min(uniq, 64)):
            b = io.BytesIO()
            Image.fromarray(host[i][..., ::-1].copy()).save(b, "JPEG", quality=80, subsampling=2)
            streams.append(b.getvalue())
        msgs = [streams[i % len(streams)] for i in range(B)]
        bufs = [fe.frames_buffer()[0] for fe in fes]

        def jpeg_rate(entropy, nthreads, feeders=1, laps=8):
            # `feeders` host threads, each driving its own share of the handles (ctypes releases the GIL inside the library: the header
            # parsing and submission of one feeder's batch overlaps the other's waits -- VERDICT r4 #4's feeder thread)
            import threading

            def jpeg_pass(nb, slots):
                inflight = []
                for k in range(nb):
                    slot = slots[k % len(slots)]
                    if len(inflight) == len(slots):
                        fes[inflight.pop(0)].wait()
                    if entropy == "gpu":                     # queued since round 6: the feeder does not wait for the decoder
                        fes[slot].decode_jpeg_batch_async(msgs, device_ptr=bufs[slot], n_threads=nthreads, for_detect=True)
                    else:
                        fes[slot].decode_jpeg_batch(msgs, n_threads=nthreads, device_ptr=bufs[slot], entropy=entropy)
                    fes[slot].submit_device(bufs[slot], B, ptrs[slot], cap, describe=True)
                    inflight.append(slot)
                while inflight:
                    fes[inflight.pop(0)].wait()
                if entropy == "gpu":
                    for slot in slots:
                        if int(np.count_nonzero(fes[slot].jpeg_status())):
                            raise RuntimeError("jpeg_ingest: frames that did not decode")

            def all_feeders(nb):
                shares = [list(range(D))[i::feeders] for i in range(feeders)]
                th = [threading.Thread(target=jpeg_pass, args=(nb // feeders, sh)) for sh in shares if sh]
                for t_ in th:
                    t_.start()
                for t_ in th:
                    t_.join()
                return (nb // feeders) * len(th)
            all_feeders(D)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            done = all_feeders(laps * D)                         # (2 * D until round 6: a sixth of such a run was the pipeline filling and draining)
            torch.cuda.synchronize()
            return done * B / (time.perf_counter() - t0)
        kb = np.mean([len(s) for s in streams]) / 1e3
        # One feeder thread since round 6: the queued decode call returns when the headers are parsed, so one thread keeps all handles fed
        # (95 - 100 k frames/s on three boxes; two feeders, the advice while the call waited for the decoder, 79 - 93 k: they contend for the
        # header-parsing threads).  two_feeders: that form, for comparison.
        nfeed = int(os.environ.get("LF_BENCH_FEEDERS", "1"))
        r_main = jpeg_rate("gpu", 8, feeders=nfeed)
        r_two = jpeg_rate("gpu", 8, feeders=2)
        sec["jpeg_ingest"] = {"value": round(r_main, 1), "unit": "frames/s", "host_threads": 8, "feeder_threads": nfeed,
                              "one_feeder": round(r_main if nfeed == 1 else jpeg_rate("gpu", 8, feeders=1), 1), "two_feeders": round(r_two, 1),
                              "what": "JPEG streams (quality 80, 4:2:0, %.0f kB each) -> lf_jpeg_decode_for_detect_async (the rows the front end reads; headers on 8 host threads; unstuffing, "
                                      "Huffman decoding by self-synchronising subsequences, DC prediction, IDCT, upsampling and colour conversion on the "
                                      "GPU) -> detect->describe->project->sanity, %d batches in flight, %d batches timed, submitted by %d feeder thread(s) "
                                      "(two_feeders: two threads with half the handles each, rounds 5's form)" % (kb, D, 8 * D, nfeed)}
        ht = max(1, min(32, (os.cpu_count() or 2) // 2))
        sec["jpeg_ingest_host_entropy"] = {"value": round(jpeg_rate("host", ht, laps=2), 1), "unit": "frames/s", "host_threads": ht,
                                           "what": "the same with lf_jpeg_decode_batch: Huffman decoding on %d host threads of a shared box (round 2's path)" % ht}
    except Exception as e:                                       # Pillow missing or similar: say so, do not fail the bench
        sec["jpeg_ingest"] = {"error": repr(e)}

    # ---- the drop-in operating point: ONE 160x120 frame through the plugin interface (default.yaml:1-2),
    # setImage + three detectLines, host arrays in and out
    det = LineDetectorHIP(dict(DEFAULT_DETECTOR_CONFIGURATION), device=device_id)
    small = [np.ascontiguousarray(host[i % uniq][::4, ::4][40:]) for i in range(32)]          # 80x160 working images
    for img in small[:4]:
        det.setImage(img)
        [det.detectLines(c) for c in ("white", "yellow", "red")]
    lat = []
    for i in range(200):
        t0 = time.perf_counter()
        det.setImage(small[i % 32])
        for c in ("white", "yellow", "red"):
            det.detectLines(c)
        lat.append(time.perf_counter() - t0)
    lat = np.sort(np.array(lat)) * 1e3
    sec["plugin_latency_160x120"] = {"median_ms": round(float(lat[100]), 3), "p90_ms": round(float(lat[180]), 3), "min_ms": round(float(lat[0]), 3),
                                     "what": "LineDetectorHIP.setImage + 3 x detectLines on one 160x80 working image (img_size [120,160], "
                                             "top_cutoff 40), host numpy in/out, 200 frames"}

    # ---- BASELINE configs[4]: associator stress, N queries x 50 000-code map (5 % planted neighbours within 40 bits)
    stress = []
    rng = np.random.default_rng(1234)
    mcodes = synth.random_codes(50000, 1234)
    am = LineAssociator(capacity=50048, color_gating=False, kept_only=False, device=device_id, tie_rule="lowest")       # the distance pass alone
    am.seed(mcodes)
    # the same map with the reference's tie rule (LF_TIE_MIHASHER: a second matrix pass that ranks the equally near entries)
    am_mih = LineAssociator(capacity=50048, color_gating=False, kept_only=False, device=device_id, tie_rule="mihasher")
    am_mih.seed(mcodes)
    for nq in (4096, 16384):
        q = synth.random_codes(nq, 99)
        planted = rng.choice(nq, nq // 20, replace=False)
        for i in planted:
            src = mcodes[rng.integers(0, 50000)].copy()
            for b in rng.choice(256, size=int(rng.integers(0, 41)), replace=False):
                src[b >> 3] ^= np.uint8(1 << (b & 7))
            q[i] = src
        dq = torch.from_numpy(q).to(dev)
        di = torch.zeros(nq, dtype=torch.int32, device=dev)
        dd = torch.zeros(nq, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        for _ in range(3):
            am.associate_device(None, dq.data_ptr(), None, nq, di.data_ptr(), dd.data_ptr())
        am.synchronize()
        am.timing()
        am.set_profiling(True)
        for _ in range(20):
            am.associate_device(None, dq.data_ptr(), None, nq, di.data_ptr(), dd.data_ptr())
        am.synchronize()
        tm = am.timing()
        am.set_profiling(False)
        ms_core = tm["assoc_mfma"][0] / max(tm["assoc_mfma"][1], 1)
        di2 = torch.zeros(nq, dtype=torch.int32, device=dev)
        for _ in range(3):
            am_mih.associate_device(None, dq.data_ptr(), None, nq, di2.data_ptr(), dd.data_ptr())
        am_mih.synchronize()
        am_mih.timing()
        am_mih.set_profiling(True)
        for _ in range(20):
            am_mih.associate_device(None, dq.data_ptr(), None, nq, di2.data_ptr(), dd.data_ptr())
        am_mih.synchronize()
        tm2 = am_mih.timing()
        am_mih.set_profiling(False)
        ms_mih = tm2["assoc_mfma"][0] / max(tm2["assoc_mfma"][1], 1)
        ops = 2.0 * nq * 50000 * 256
        stress.append({"N": nq, "M": 50000, "assoc_ms": round(ms_core, 4),
                       "Pop_per_s": round(ops / (ms_core * 1e-3) / 1e15, 3),
                       "frac_of_int8_mfma_peak": round(ops / (ms_core * 1e-3) / 1e15 / INT8_MFMA_PEAK_POPS, 3),
                       "frac_of_fp4_mfma_peak": round(ops / (ms_core * 1e-3) / 1e15 / (2 * INT8_MFMA_PEAK_POPS), 3),
                       "matched_within_128": int((di >= 0).sum().item()),
                       "assoc_ms_tie_rule_mihasher": round(ms_mih, 4), "tie_pass_ms": round(ms_mih - ms_core, 4),
                       "indices_changed_by_the_tie_rule": int((di != di2).sum().item())})
    am.close()
    am_mih.close()
    sec["assoc_stress_configs4"] = {"rows": stress, "peak_Pop_per_s": INT8_MFMA_PEAK_POPS,
                                    "what": "lf_map_associate on a 50 000-code map kept packed on the device; ops = 2*N*M*256 (SURVEY 8d); the ungated map runs on the FP4 matrix "
                                            "instruction (e2m1 +-1 operands, exact f32 accumulation), whose dense peak is 2 x the int8 peak: both fractions are given; "
                                            "assoc_ms = the whole association (ONE launch: query expansion, MFMA loop, merge, report), HIP events on the map's stream, 20 calls; "
                                            "assoc_ms_tie_rule_mihasher = the same with lf_map_set_tie_rule(LF_TIE_MIHASHER): + the pass that ranks equally near entries "
                                            "by the reference's discovery order (k_assoc_ties.hip)"}
    return sec


if __name__ == "__main__":
    main()
