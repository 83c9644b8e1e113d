cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export REPS=3
python3 $R/tools/pt_layer_bench.py > $R/gpurun_out/pt_bench.log 2>&1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
i=$((i+1))
rocprofv3 --pmc $set --kernel-trace -d /tmp/p/pt$i -o pt -- python3 $R/tools/pt_layer_bench.py > $R/gpurun_out/pt_pmc$i.log 2>&1
python3 $R/tools/rocpd_pmc.py $(find /tmp/p/pt$i -name "*.db" | head -1) 400 | grep -E "flm?::k_[pb]" > $R/gpurun_out/pt_pmc$i.txt
done
