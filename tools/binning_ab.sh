for V in "" seg seg3 radix; do
  DAS3R_BINNING=$V python bench.py --workload c4 --steps 100 --warmup 20 --no-extras --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('binning=%-6s step_ms=%.4f fwd_only=%.4f' % ('${V:-auto}', d['ms_per_step'], d['fwd_only']['ms_per_step']), {n:v['ms_per_step'] for n,v in k.items()})"
done
