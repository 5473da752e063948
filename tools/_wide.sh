export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_trainstep.py -m gpu -x -q -k "not schedule" 2>&1 | tail -8
F="--fused-adam --fused-loss --fused-pre --breakdown"
for d in smooth noise; do
  echo "## $d"; timeout 200 python tools/train_bench.py $F --depth $d 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j.get('train_step_ms'), j.get('kernel_launches_per_step'))
for k,v in list(j.get('breakdown_ms',{}).items())[:12]: print('   %-62s %.3f'%(k[:60],v))"
done
