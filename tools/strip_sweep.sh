# Traffic and time of the compositing kernels against the strip height of the tile order (run on the GPU box): bash tools/strip_sweep.sh "8 16 32"
for S in ${1:-0 2 4 8}; do
  DAS3R_TILE_STRIP=$S timeout 200 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('strip $S c4', d['ms_per_step'], {k:(v['ms_per_step'], v.get('traffic_over_alg')) for k,v in d['kernels'].items() if k.startswith('render')})"
done
