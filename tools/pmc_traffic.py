#!/usr/bin/env python3
"""Derive per-launch HBM traffic from two rocprofv3 PMC passes (WRITE_SIZE, FETCH_SIZE; separate runs).

usage: tools/pmc_traffic.py <pmc_write.csv> <pmc_fetch.csv> <tag> > profiles/traffic.json

Counters are in KB; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (the counter
reports half the bytes of wide coalesced reads).  Output keys: kernel short names plus the stage
aliases bench.py looks up for `roofline.traffic`."""
import csv, hashlib, json, os, re, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def digest(rel):
    with open(os.path.join(ROOT, rel), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]

def per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            m = re.match(r"(?:void )?(?:lf::)?([A-Za-z0-9_]+)", r["Kernel_Name"])
            name = m.group(1) if m else r["Kernel_Name"]
            if not name.startswith("k_"):
                continue
            tot[name] += float(r["Counter_Value"]); cnt[name] += 1
    return {k: tot[k] / cnt[k] for k in tot}

w = per_kernel(sys.argv[1], "WRITE_SIZE")
f = per_kernel(sys.argv[2], "FETCH_SIZE")
tag = sys.argv[3]
tr = {k: int(round((w.get(k, 0.0) + 2.0 * f.get(k, 0.0)) * 1024)) for k in sorted(set(w) | set(f))}
alias = {"pre(resize+correct+hsv+masks+dilate)": "k_pre", "canny_nms": "k_canny_nms",
         "lbd_gray_blur_sobel": "k_lbd_grad", "canny_hysteresis": "k_hysteresis"}
for a, k in alias.items():
    if k in tr:
        tr[a] = tr[k]
json.dump({"workload": {"batch": 256, "geometry": "fullres"},
           # bench.py quotes roofline.traffic only while the dominant kernel's source is the one measured here
           "source_digest": {"k_pre": digest("lane_slam_amd/csrc/k_pre.hip")},
           "source": "rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE in separate passes of `bench.py --steps 2 --warmup 1 "
                     "--cpu-frames -1 --depth 1` (profiles/%s_pmc_*.csv); counters are in KB; FETCH_SIZE doubled per "
                     "MI355X_MICROARCH.md (gfx950 reports half the bytes of coalesced reads; the factor was calibrated on this "
                     "pipeline's own access shapes -- 4, 8, 12 and 16 bytes per lane all read x2.000, profiles/r02_fetch_probe.json); "
                     "derived by tools/pmc_traffic.py" % tag,
           "traffic_bytes_per_launch": tr}, sys.stdout, indent=1)
