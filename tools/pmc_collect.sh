#!/bin/bash
# usage: tools/pmc_collect.sh <workload> <outdir>   (run on the GPU box)
# Separate rocprofv3 --pmc passes (kernel trace only, nothing else traced) over four steps of the workload (bench.py --pmc-child:
# the plain step loop, no event bracketing), every pass under its own 90 s limit; then the per-kernel summary.
W=${1:-c4}; OUT=${2:-gpurun_out/pmc_$W}; R=$PWD
export PYTHONPATH=$R; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 ${PMC_LIMIT:-90} rocprofv3 --pmc $set --kernel-trace -d $R/$OUT/pass$i -o pmc --output-format csv -- \
      python $R/bench.py --pmc-child --workload $W --steps 3 ${EXTRA} > /dev/null 2>&1 || echo "pass $i ($set): rc $?"
done
cd $R; python tools/pmc_summary.py $OUT
