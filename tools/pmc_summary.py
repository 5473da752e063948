#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes written by tools/pmc_collect.sh: per kernel, mean counter value per launch.
FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3; on gfx950 FETCH_SIZE under-counts wide coalesced reads by 2x
(/opt/skills/guides/MI355X_MICROARCH.md §HBM) — both the raw and the corrected figure are printed.
    python tools/pmc_summary.py gpurun_out/pmc_c4 [--json profiles/pmc_c4.json]"""
import collections
import csv
import glob
import json
import sys


def main():
    d = sys.argv[1]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{d}/pass*/**/*counter_collection.csv", recursive=True) + glob.glob(f"{d}/pass*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("das3r::", "").split("<")[0]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, cs in sorted(agg.items()):
        if k.startswith("at::") or "elementwise" in k or "Memset" in k:
            continue
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        ent = dict(m)
        if "FETCH_SIZE" in m or "WRITE_SIZE" in m:
            f, w = m.get("FETCH_SIZE", 0.0) * 1024, m.get("WRITE_SIZE", 0.0) * 1024
            ent["hbm_bytes_per_launch_raw"] = f + w
            ent["hbm_bytes_per_launch"] = 2 * f + w   # gfx950 FETCH_SIZE correction for wide streaming reads
        out[k] = ent
        print(k)
        for c, v in sorted(ent.items()):
            print(f"    {c:28s} {v:.4e}")
    if "--json" in sys.argv:
        json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
