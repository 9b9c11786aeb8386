#!/usr/bin/env python3
"""Copy the summaries of one tools/profile_round.sh run from gpurun_out/<tag>/ into profiles/<prefix>_* (tracked).
usage: python tools/collect_profiles.py r02c r02_c"""
import csv, glob, os, shutil, subprocess, sys
tag, prefix = sys.argv[1], sys.argv[2]
T = os.path.join("gpurun_out", tag)
P = "profiles"

def one(pattern):
    f = glob.glob(os.path.join(T, pattern), recursive=True)
    assert f, pattern
    return f[0]

shutil.copy(os.path.join(T, "bench_n1.json"), os.path.join(P, prefix + "_bench_n1.json"))
if os.path.exists(os.path.join(T, "bench_driver_form.json")):
    shutil.copy(os.path.join(T, "bench_driver_form.json"), os.path.join(P, prefix + "_bench_driver_form.json"))
shutil.copy(os.path.join(T, "bench_d6_rocprof.json"), os.path.join(P, prefix + "_bench_depth6_under_rocprof.json"))
shutil.copy(os.path.join(T, "bench_d1_rocprof.json"), os.path.join(P, prefix + "_bench_depth1_under_rocprof.json"))
shutil.copy(one("d6/**/*kernel_stats.csv"), os.path.join(P, prefix + "_kernel_stats_depth6.csv"))
shutil.copy(one("d1/**/*kernel_stats.csv"), os.path.join(P, prefix + "_kernel_stats_depth1.csv"))
shutil.copy(one("as/**/*kernel_stats.csv"), os.path.join(P, prefix + "_assoc_kernel_stats.csv"))
with open(os.path.join(T, "assoc_rate.txt")) as f, open(os.path.join(P, prefix + "_assoc_rate.txt"), "w") as g:
    g.writelines(l for l in f if "amdgpu.ids" not in l)

def slim(src, dst):
    """counter rows of the lanefront kernels only, the columns that matter"""
    cols = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value"]
    with open(src) as f, open(dst, "w", newline="") as g:
        w = csv.DictWriter(g, fieldnames=cols)
        w.writeheader()
        for r in csv.DictReader(f):
            if "lf::" not in r["Kernel_Name"]:
                continue
            r2 = {k: r[k] for k in cols}
            r2["Kernel_Name"] = r2["Kernel_Name"].split("(")[0]
            w.writerow(r2)

slim(one("pw/**/*counter_collection.csv"), os.path.join(P, prefix + "_pmc_write_size.csv"))
slim(one("pf/**/*counter_collection.csv"), os.path.join(P, prefix + "_pmc_fetch_size.csv"))
slim(one("pa/**/*counter_collection.csv"), os.path.join(P, prefix + "_pmc_assoc_sq.csv"))
slim(one("pg/**/*counter_collection.csv"), os.path.join(P, prefix + "_pmc_assoc_grbm.csv"))
for sub, name in (("pq", "_pmc_grow_sq.csv"), ("pq2", "_pmc_grow_sq2.csv"), ("pq6", "_pmc_grow_sq_d6.csv"), ("pq26", "_pmc_grow_sq2_d6.csv")):
    f = glob.glob(os.path.join(T, sub + "/**/*counter_collection.csv"), recursive=True)
    if f:
        slim(f[0], os.path.join(P, prefix + name))
with open(os.path.join(P, "traffic.json"), "w") as g:
    subprocess.check_call([sys.executable, "tools/pmc_traffic.py", os.path.join(P, prefix + "_pmc_write_size.csv"),
                           os.path.join(P, prefix + "_pmc_fetch_size.csv"), prefix], stdout=g)
subprocess.check_call([sys.executable, "tools/pmc_grow.py", prefix])
# round 5: the A/B configurations, the 1080p geometry, the seed-order kernel's phase stamps, instruction counts per kernel
for f in glob.glob(os.path.join(T, "bench_ab_*.json")) + glob.glob(os.path.join(T, "bench_hd.json")):
    shutil.copy(f, os.path.join(P, prefix + "_" + os.path.basename(f)))
if os.path.exists(os.path.join(T, "seed_stamps.txt")):
    with open(os.path.join(T, "seed_stamps.txt")) as f, open(os.path.join(P, prefix + "_seed32_stamps.txt"), "w") as g:
        g.writelines(l for l in f if l.startswith("[seed32"))
for name, dst in (("whatif.txt", "_whatif.txt"), ("handle_footprint.txt", "_handle_footprint.txt"), ("content_whatif.txt", "_content_whatif.txt"),
                  ("grow_stamps_SYNTHETIC.txt", "_grow_stamps_synthetic.txt"), ("grow_stamps_REAL.txt", "_grow_stamps_real.txt"), ("grow_stamps_CLUTTER.txt", "_grow_stamps_clutter.txt")):
    if os.path.exists(os.path.join(T, name)):
        with open(os.path.join(T, name)) as f, open(os.path.join(P, prefix + dst), "w") as g:
            g.writelines(l for l in f if "amdgpu.ids" not in l)
f = glob.glob(os.path.join(T, "pi/**/*counter_collection.csv"), recursive=True)
if f:
    tot, cnt = {}, {}
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0].split("::")[-1]
        tot.setdefault(k, {}).setdefault(r["Counter_Name"], 0.0)
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_INSTS_VALU":
            cnt[k] = cnt.get(k, 0) + 1
    with open(os.path.join(P, prefix + "_instructions_per_kernel.txt"), "w") as g:
        g.write("wave instructions per launch (rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES; bench.py --depth 1, 256-frame batch)\n")
        for k, v in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
            n = max(cnt.get(k, 1), 1)
            if v.get("SQ_INSTS_VALU", 0) / n < 1e4:
                continue
            g.write("%-32s VALU %8.2f M  SALU %8.2f M  LDS %7.2f M  wave quad-cycles %9.1f M\n" % (k[:32], v.get("SQ_INSTS_VALU", 0) / n / 1e6, v.get("SQ_INSTS_SALU", 0) / n / 1e6,
                                                                                                 v.get("SQ_INSTS_LDS", 0) / n / 1e6, v.get("SQ_WAVE_CYCLES", 0) / n / 1e6))
# round 6: EDLines / KeyLines and JPEG ingest
for sub, name in (("kl", "_keylines_kernel_stats.csv"), ("jp", "_ingest_kernel_stats.csv")):
    f = glob.glob(os.path.join(T, sub + "/**/*kernel_stats.csv"), recursive=True)
    if f:
        shutil.copy(f[0], os.path.join(P, prefix + name))
for name in ("keylines_rate.txt", "ingest_rate.txt", "jh_passes.txt", "ed_stamps.txt"):
    if os.path.exists(os.path.join(T, name)):
        with open(os.path.join(T, name)) as f, open(os.path.join(P, prefix + "_" + name), "w") as g:
            g.writelines(l for l in f if "amdgpu.ids" not in l and "simple_timer" not in l)
print("collected", prefix)
