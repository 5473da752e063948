export PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_gpu_raster.py -m gpu -x -q -k "forward_kernels_agree" 2>&1 | tail -15
F="--fused-adam --fused-loss --fused-pre --breakdown"
for d in smooth noise; do for r in rows lanes rows lanes; do
  echo "## $d $r"; DAS3R_RENDER=$r timeout 200 python tools/train_bench.py $F --depth $d 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j.get('train_step_ms'), {k[:45]:v for k,v in j.get('breakdown_ms',{}).items() if 'render_' in k})"
done; done
for w in ds dsc c4; do for r in rows lanes; do echo "## $w $r"; DAS3R_RENDER=$r timeout 120 python tools/gpu_perf.py --workloads $w --steps 30 2>&1 | grep "==\|render_forward"; done; done
