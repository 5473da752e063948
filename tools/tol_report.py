#!/usr/bin/env python3
"""How far under the parity bars the kernels are.  On the GPU box:
    DAS3R_TOL_REPORT=$PWD/gpurun_out/tol.jsonl python -m pytest tests -m gpu -q ; python tools/tol_report.py gpurun_out/tol.jsonl
tests/util.py appends one line per gradient check: the measured max-norm error and its bar, the number of elements beyond the bar
(threshold flips at full size), and the fraction of elements that would miss element-wise bars 1x / 10x / 100x tighter."""
import collections
import json
import re
import sys

rows = [json.loads(line) for line in open(sys.argv[1])]
test_of = lambda r: re.sub(r"\[.*| \(call\)", "", r["test"].split("::")[-1])
print(f"{len(rows)} records")
worst = collections.defaultdict(lambda: [0.0, 0.0, 0, 0])
for r in rows:
    k = (r["test"].split("::")[0].split("/")[-1], test_of(r))
    if r["kind"] == "max_norm":
        worst[k][0] = max(worst[k][0], r["value"])
        worst[k][1] = max(worst[k][1], r["tol"])
    elif r["kind"] == "over_bar":
        if r["value"] > worst[k][2]:
            worst[k][2], worst[k][3] = int(r["value"]), int(r["tol"])
print("worst max-norm error per test (bar; most elements beyond the bar in one tensor / its size):")
for k, (v, t, n, size) in sorted(worst.items(), key=lambda kv: -kv[1][0]):
    print(f"  {v:9.3e}  (bar {t:.0e}; {n} / {size})  {k[0]}::{k[1]}")
for d in ("1", "10", "100"):
    el = [r for r in rows if r["kind"] == "elementwise/" + d]
    if not el:
        continue
    el.sort(key=lambda r: -r["value"])
    print(f"element-wise bars / {d}: {len(el)} checks, {sum(r['value'] > 0 for r in el)} with any element beyond; worst five:")
    for r in el[:5]:
        print(f"  {r['value']:9.3e} of the elements  {r['what'][:48]:48s} {test_of(r)}")
