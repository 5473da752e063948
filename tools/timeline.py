#!/usr/bin/env python3
"""Per-step GPU timeline from a rocprofv3 --kernel-trace (+ --memory-copy-trace) CSV dump: kernel durations and the idle gap in
front of each launch, averaged over the steps found.   python tools/timeline.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("das3r::", "")))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY_" + r.get("Direction", "?")))
rows.sort()
# steps start at preprocess_kernel
steps, cur = [], []
for r in rows:
    if r[2].startswith("preprocess_kernel") and cur:
        steps.append(cur)
        cur = []
    cur.append(r)
steps.append(cur)
steps = [s for s in steps if any(x[2].startswith("render_backward") for x in s)]
steps = steps[len(steps) // 2:]          # steady state
n = len(steps)
L = max(len(s) for s in steps)
print(f"{n} steady-state steps")
acc = defaultdict(lambda: [0.0, 0.0, 0])
for s in steps:
    prev_end = None
    for i, (a, b, name) in enumerate(s):
        key = (i, name)
        acc[key][0] += (b - a) / 1e3
        acc[key][1] += ((a - prev_end) / 1e3) if prev_end is not None else 0.0
        acc[key][2] += 1
        prev_end = max(prev_end or b, b)
tot_k = tot_g = 0.0
for (i, name), (dur, gap, c) in sorted(acc.items()):
    print(f"{i:3d} {name[:44]:44s} dur {dur / c:8.1f} us   gap before {gap / c:8.1f} us   (seen {c}/{n})")
    tot_k += dur / n
    tot_g += gap / n
print(f"sum kernels {tot_k:.1f} us, sum gaps {tot_g:.1f} us")
