#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV of a pipelined run (tools/pipe_content.py) and says, for the kernels whose name contains
KEY (default lsd_grow): how long they last, how many of them run side by side, and how much of the time at least one is running.
    python3 tools/pipe_overlap.py <kernel_trace.csv> [KEY]"""
import csv, sys
key = sys.argv[2] if len(sys.argv) > 2 else "lsd_grow"
rows = []
allrows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        b, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        allrows.append((b, e, r["Kernel_Name"]))
        if key in r["Kernel_Name"]:
            rows.append((b, e, r["Kernel_Name"]))
rows.sort()
# the steady part: drop the first and last quarter of the launches
q = len(rows) // 4
mid = rows[q:len(rows) - q]
t0, t1 = mid[0][0], mid[-1][1]
dur = sorted((e - b) / 1e6 for b, e, _ in mid)
print("%d kernels matching %r, %d in the middle half; duration ms: min %.2f median %.2f p90 %.2f max %.2f" % (
    len(rows), key, len(mid), dur[0], dur[len(dur) // 2], dur[int(len(dur) * 0.9)], dur[-1]))
ev = []
for b, e, _ in rows:
    ev.append((b, 1)); ev.append((e, -1))
ev.sort()
cur, last, area, busy = 0, None, 0.0, 0.0
hist = {}
for t, dlt in ev:
    if last is not None and t > t0 and last < t1:
        a, z = max(last, t0), min(t, t1)
        if z > a:
            area += cur * (z - a)
            hist[cur] = hist.get(cur, 0) + (z - a)
            if cur > 0: busy += z - a
    cur += dlt
    last = t
span = t1 - t0
print("between %.1f and %.1f ms: %.2f running side by side on average, at least one %.1f %% of the time; start to start %.3f ms" % (
    0.0, span / 1e6, area / span, 100.0 * busy / span, span / 1e6 / max(len(mid) - 1, 1)))
print("side by side: " + "  ".join("%d: %.0f%%" % (k, 100.0 * v / span) for k, v in sorted(hist.items())))
# everything else in the same window
other = {}
for b, e, n in allrows:
    if b >= t0 and e <= t1 and key not in n:
        nm = n.split("(")[0][-40:]
        o = other.setdefault(nm, [0, 0.0]); o[0] += 1; o[1] += (e - b) / 1e6
print("other kernels in the window (count, total ms, mean ms):")
for nm, (c_, t_) in sorted(other.items(), key=lambda kv: -kv[1][1])[:12]:
    print("   %-42s %5d %9.2f %8.3f" % (nm, c_, t_, t_ / c_))
