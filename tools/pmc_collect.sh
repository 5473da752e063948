#!/bin/bash
# usage: tools/pmc_collect.sh <workload> <outdir>   (run on the GPU box; separate rocprofv3 --pmc passes, no other tracing)
W=${1:-c4}; OUT=${2:-gpurun_out/pmc_$W}; R=$PWD
export PYTHONPATH=$R; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $R/$OUT/pass$i -o pmc --output-format csv -- python $R/tools/gpu_perf.py --workloads $W --steps 2 ${EXTRA} > /dev/null 2>&1
done
cd $R; python tools/pmc_summary.py $OUT
