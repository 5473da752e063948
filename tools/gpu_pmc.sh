# usage: tools/gpu_pmc.sh <workload> <DAS3R_RENDER_BWD value> <tag>     (run on the GPU box)
W=$1; V=$2; TAG=$3; R=$PWD; export PYTHONPATH=$R; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
export DAS3R_RENDER_BWD=$V
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_LEVEL_LDS" \
           "SQ_INST_LEVEL_VMEM SQ_WAVE_DEP_WAIT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_ACTIVE_INST_EXP_GDS SQ_INSTS_WAVE32_LDS"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc_$TAG/pass$i -o pmc --output-format csv -- python $R/tools/gpu_perf.py --workloads $W --steps 2 > $R/gpurun_out/pmc_${TAG}_pass$i.log 2>&1
done
unset DAS3R_RENDER_BWD
cd $R
(python tools/pmc_summary.py gpurun_out/pmc_$TAG | grep -A34 "^render_backward" | head -36) > gpurun_out/pmc_$TAG.txt 2>&1
cat gpurun_out/pmc_$TAG.txt
