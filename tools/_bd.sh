export PYTHONPATH=$PWD
F="--fused-adam --fused-loss --fused-pre --breakdown"
for d in smooth noise; do
  echo "## $d"; timeout 200 python tools/train_bench.py $F --depth $d 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j.get('train_step_ms'), j.get('kernel_launches_per_step'), j.get('device_ms_per_step'))
for k,v in j.get('breakdown_ms',{}).items(): print('   %-62s %.3f'%(k[:60],v))"
done
