# PMC passes over the pointops2 window-attention probe (tools/pointops2_bench.py); usage on the GPU box: bash tools/prof_p2.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32"; do
i=$((i+1))
rocprofv3 --pmc $set --kernel-trace -d /tmp/p/p2$i -o p2 -- python3 $R/tools/pointops2_bench.py > $R/gpurun_out/p2_pmc$i.log 2>&1
python3 $R/tools/rocpd_pmc.py $(find /tmp/p/p2$i -name "*.db" | head -1) 400 | grep -E "k_dot3|k_step" > $R/gpurun_out/p2_pmc$i.txt
done
