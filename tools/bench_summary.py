#!/usr/bin/env python3
"""Print the key fields of bench.py JSON lines read from stdin (one per line)."""
import json
import sys
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d.get("roofline") or {}
    c = d.get("cpu_baseline") or {}
    print(f"{d['config']['name']} K={d['steps']} W={d['warmup']} value={d['value']} {d['unit']} ms/step={d['ms_per_step']} "
          f"I={d['config']['num_rendered']} dom={r.get('kernel')} {r.get('achieved')} GB/s frac={r.get('frac')} "
          f"pipeline={r.get('pipeline', {}).get('GBps')} cpu={c.get('value')}")
