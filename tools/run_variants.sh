#!/bin/bash
# on the GPU box: per-kernel times of each library variant (tools/build_variants.sh), same box, back to back
#   bash tools/run_variants.sh "c4,ds" base A B C base
W=$1; shift
cp das3r_amd/libdas3r_hip.so /tmp/libdas3r_hip.default.so
for v in "$@"; do
  if [ "$v" = default ]; then cp /tmp/libdas3r_hip.default.so das3r_amd/libdas3r_hip.so; else cp das3r_amd/libdas3r_hip.$v.so das3r_amd/libdas3r_hip.so; fi
  echo "#### variant $v"
  timeout 120 python tools/gpu_perf.py --workloads $W --steps 30 2>&1 | grep "==\|render_\|preprocess\|onesweep\|scan_emit"
done
cp /tmp/libdas3r_hip.default.so das3r_amd/libdas3r_hip.so
