#!/usr/bin/env python3
"""Calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE for the access shapes this pipeline uses (MI355X_MICROARCH.md, HBM
section: 16 B/lane streams read exactly half their bytes in FETCH_SIZE on gfx950; "other access widths are
uncalibrated: calibrate on a known byte count in your own access pattern").

    run:      rocprofv3 --pmc FETCH_SIZE --output-format csv -d OUT -- python3 tools/fetch_probe.py run
              rocprofv3 --pmc WRITE_SIZE --output-format csv -d OUT2 -- python3 tools/fetch_probe.py run
    analyse:  python3 tools/fetch_probe.py analyse OUT/.../*counter_collection.csv [OUT2/...csv] > profiles/r02_fetch_probe.json

Every probe kernel streams a 1 GiB buffer (4x the Infinity Cache) once per launch; counters are in KB.
"""
import csv, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BYTES = 1 << 30

if sys.argv[1] == "run":
    from lane_slam_amd import FrontEnd, default_config
    fe = FrontEnd(default_config("parity"))
    for width, write in ((4, 0), (8, 0), (12, 0), (16, 0), (43, 0), (4, 1), (16, 1)):
        rc = fe.lib.lf_debug_probe(fe.h, width, write, BYTES, 3)
        assert rc == 0, rc
    fe.close()
else:
    out = {"bytes_per_launch": BYTES, "counters_in": "KB", "probes": {}}
    for path in sys.argv[2:]:
        with open(path) as f:
            for r in csv.DictReader(f):
                m = re.search(r"k_probe_(read|write)<(\d+)>", r["Kernel_Name"])
                if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE") or not (m or "k_probe_stencil" in r["Kernel_Name"]):
                    continue
                key = "%s_%sB_per_lane" % (m.group(1), m.group(2)) if m else "read_stencil3_buffer_loads(k_canny_nms)"
                e = out["probes"].setdefault(key, {})
                e.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for key, e in out["probes"].items():
        for c in list(e):
            vals = e[c]
            mean = sum(vals) / len(vals)
            e[c] = {"mean_KB": round(mean, 1), "launches": len(vals), "bytes_per_counted_byte": round(BYTES / (mean * 1024), 4) if mean else None}
    json.dump(out, sys.stdout, indent=1)
