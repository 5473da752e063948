#!/bin/bash
# One bounded rocprofv3 --pmc pass over a few steps of a workload (run on the GPU box).
#   tools/pmc_pass.sh <workload> <tag> <counter> [<counter> ...]      -> gpurun_out/pmc_<tag>/  + summary on stdout
# Every pass is its own rocprofv3 run with the kernel trace only, under a 90 s limit (a pass that hangs costs 90 s, not the call).
W=$1; TAG=$2; shift 2; R=$PWD
export PYTHONPATH=$R TMPDIR=/tmp; mkdir -p $R/gpurun_out/pmc_$TAG; cd /tmp
timeout -k 5 ${PMC_LIMIT:-90} rocprofv3 --pmc "$@" --kernel-trace -d $R/gpurun_out/pmc_$TAG/pass1 -o pmc --output-format csv -- \
    python $R/bench.py --pmc-child --workload $W --steps 3 > $R/gpurun_out/pmc_$TAG.log 2>&1
echo "pass rc=$?"
cd $R; python tools/pmc_summary.py gpurun_out/pmc_$TAG --json gpurun_out/pmc_$TAG.json
