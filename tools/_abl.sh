export PYTHONPATH=$PWD
cp das3r_amd/libdas3r_hip.so /tmp/default.so
cp das3r_amd/libdas3r_hip.exp.so das3r_amd/libdas3r_hip.so
F="--fused-adam --fused-loss --fused-pre --breakdown"
for d in smooth noise; do
 for cfg in "blk192 -" "blk128p1 0" "blk128p1 1" "blk128p1 2" "blk128p1 4" "blk128p1 8"; do
  set -- $cfg
  echo "## $d bwd=$1 ablate=$2"
  if [ "$2" = "-" ]; then unset DAS3R_ABLATE; else export DAS3R_ABLATE=$2; fi
  DAS3R_RENDER_BWD=$1 timeout 200 python tools/train_bench.py $F --depth $d --iters 20 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j.get('train_step_ms'), {k[:48]:v for k,v in j.get('breakdown_ms',{}).items() if 'render_backward' in k})"
 done
done
cp /tmp/default.so das3r_amd/libdas3r_hip.so
