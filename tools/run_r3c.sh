export DAS3R_RENDER_BWD=blk128
bash tools/pmc_pass.sh c4 blk_a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR > /dev/null
bash tools/pmc_pass.sh c4 blk_b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA > /dev/null
bash tools/pmc_pass.sh c4 blk_c GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_LEVEL_LDS SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU > /dev/null
python - <<'PY'
import json
for t in "abc":
    try:
        d=json.load(open(f"gpurun_out/pmc_blk_{t}.json"))
        for k in d:
            if "render" in k: print(k, {c: f"{v:.3e}" for c,v in d[k].items()})
    except Exception as e: print(t, e)
PY
