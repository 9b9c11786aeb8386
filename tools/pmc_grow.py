#!/usr/bin/env python3
"""Wave-slot occupancy and issue utilisation of k_lsd_grow from rocprofv3 SQ counter passes ("make
latency bound a number").

usage: tools/pmc_grow.py <prefix>        e.g. r03_b  ->  reads profiles/<prefix>_pmc_grow_sq*.csv (+ _d6 variants),
                                                         writes profiles/grow_counters.json

Counter units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count QUAD-cycles summed over all waves;
SQ_BUSY_CYCLES counts cycles per shader engine (32 on the MI355X); SQ_INSTS_* count wave instructions.  WAIT_ANY +
WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES (disjoint)."""
import csv, hashlib, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SE, N_SIMD, SLOTS_AT_80_VGPR = 32, 1024, 5120          # shader engines, SIMDs, wave slots at k_lsd_grow's 96 VGPRs (5 per SIMD since round 4; 80 / 6 before)


def digest(*rel):
    h = hashlib.sha256()
    for r in rel:
        with open(os.path.join(ROOT, r), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def per_kernel(path, kernel):
    tot, cnt = defaultdict(float), defaultdict(int)
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Kernel_Name"].split("::")[-1].split("(")[0].strip() != kernel:
                continue
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
    return {k: tot[k] / cnt[k] for k in tot}


def derive(c):
    if not c or "SQ_WAVE_CYCLES" not in c:
        return None
    wave_cycles = 4.0 * c["SQ_WAVE_CYCLES"]
    kernel_cycles = c["SQ_BUSY_CYCLES"] / N_SE if "SQ_BUSY_CYCLES" in c else None
    out = {"waves": int(c.get("SQ_WAVES", 0)), "wave_cycles": wave_cycles, "kernel_busy_cycles": kernel_cycles}
    if kernel_cycles:
        out["avg_resident_waves"] = round(wave_cycles / kernel_cycles, 1)
        out["wave_slot_occupancy"] = round(wave_cycles / kernel_cycles / SLOTS_AT_80_VGPR, 4)
        if "SQ_INSTS_VALU" in c:
            out["valu_instructions"] = c["SQ_INSTS_VALU"]
            out["valu_issue_utilisation"] = round(4.0 * c["SQ_INSTS_VALU"] / (N_SIMD * kernel_cycles), 4)     # 4 cycles per wave64 instruction and SIMD
    for name, key in (("frac_wave_cycles_parked(s_waitcnt/barrier/sleep)", "SQ_WAIT_ANY"), ("frac_wave_cycles_issue_stalled", "SQ_WAIT_INST_ANY"),
                      ("frac_wave_cycles_issuing", "SQ_ACTIVE_INST_ANY"), ("frac_wave_cycles_issuing_valu", "SQ_ACTIVE_INST_VALU")):
        if key in c:
            out[name] = round(4.0 * c[key] / wave_cycles, 4)
    return out


prefix = sys.argv[1]
P = os.path.join(ROOT, "profiles")
res = {"kernel": "k_lsd_grow", "source_digest": digest("lane_slam_amd/csrc/lsd_grow.h", "lane_slam_amd/csrc/k_lsd_grow.hip"),
       "source": "rocprofv3 --pmc passes of bench.py (tools/profile_round.sh), profiles/%s_pmc_grow_sq*.csv; units per MI355X_MICROARCH.md" % prefix}
for tag, suffix in (("one_batch_in_flight", ""), ("six_batches_in_flight", "_d6")):
    c = {}
    for part in ("sq", "sq2"):
        path = os.path.join(P, "%s_pmc_grow_%s%s.csv" % (prefix, part, suffix))
        # the instance that does the work on the synthetic frames: <0> (problems that fit the LDS slice; <1>, launched behind it,
        # finds nothing to do there); earlier sets: one kernel, or two templated on a bool
        for name in ("k_lsd_grow_bm", "k_lsd_grow<0>", "k_lsd_grow", "k_lsd_grow<false>"):
            got = per_kernel(path, name)
            if got:
                c.update(got)
                break
    d = derive(c)
    if d:
        res[tag] = d
json.dump(res, open(os.path.join(P, "grow_counters.json"), "w"), indent=1)
print(json.dumps(res, indent=1))

# The streaming kernels below the 40 % HBM line: what bounds them, from the same passes.  valu_issue_utilisation
# near 1 = every SIMD issues a vector instruction almost every cycle: the kernel is bound by its instruction count, and its
# HBM fraction can only rise by executing fewer instructions per byte.
STREAMING = {"k_pre": ["lane_slam_amd/csrc/k_pre.hip"], "k_canny_nms": ["lane_slam_amd/csrc/k_canny.hip"],
             "k_lbd_grad": ["lane_slam_amd/csrc/k_lbd.hip"], "k_lsd_grad": ["lane_slam_amd/csrc/k_lsd_grad.hip"],
             "k_lbd": ["lane_slam_amd/csrc/k_lbd.hip"], "k_hysteresis_cols<32, 256>": ["lane_slam_amd/csrc/k_canny.hip"], "k_lsd_order_bm": ["lane_slam_amd/csrc/k_lsd_order.hip"]}
out = {"source": res["source"], "kernels": {}}
for kname, files in STREAMING.items():
    c = {}
    for part in ("sq", "sq2"):
        c.update(per_kernel(os.path.join(P, "%s_pmc_grow_%s.csv" % (prefix, part)), kname))
    d = derive(c)
    if d:
        d.pop("wave_slot_occupancy", None)          # that figure is priced at k_lsd_grow's register budget
        d["source_digest"] = digest(*files)
        out["kernels"][kname] = d
json.dump(out, open(os.path.join(P, "kernel_counters.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
