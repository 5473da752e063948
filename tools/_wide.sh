export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_gpu_raster.py -m gpu -x -q -k "forward_kernels_agree" 2>&1 | tail -5
F="--fused-adam --fused-loss --fused-pre --breakdown"
for d in smooth noise smooth noise; do
  echo "## $d"; timeout 200 python tools/train_bench.py $F --depth $d 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j.get('train_step_ms'), {k[:45]:v for k,v in j.get('breakdown_ms',{}).items() if 'render_' in k})"
done
