export PYTHONPATH=$PWD
cp das3r_amd/libdas3r_hip.so /tmp/default.so
F="--fused-adam --fused-loss --fused-pre --breakdown"
for v in "$@"; do
  if [ "$v" = default ]; then cp /tmp/default.so das3r_amd/libdas3r_hip.so; else cp das3r_amd/libdas3r_hip.$v.so das3r_amd/libdas3r_hip.so || exit 1; fi
  for d in smooth noise; do
  echo "## $v $d"; timeout 200 python tools/train_bench.py $F --depth $d 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j.get('train_step_ms'), {k[:45]:v for k,v in j.get('breakdown_ms',{}).items() if 'render_' in k})"
  done
done
cp /tmp/default.so das3r_amd/libdas3r_hip.so
