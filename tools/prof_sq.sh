# SQ counter passes for the hot main-stream kernels of the headline step (what limits each kernel: parked / issue-stalled / busy,
# LDS conflicts, MFMA busy, occupancy); usage on the GPU box: bash tools/prof_sq.sh <tag> [extra bench args]
# Three passes of 8 SQ counters each (separate from --stats; bench.py --throttle because rocprofv3 --pmc serialises kernels).
TAG=${1:-r04_a}; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|GRBM|TCC|TCP|TA|TD)_[A-Z0-9_]+" | sort -u > $R/gpurun_out/${TAG}_counters_available.txt
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" \
           "SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
i=$((i+1))
for attempt in 1 2 3; do
rm -rf /tmp/p/sq$i
timeout -s KILL 300 rocprofv3 --pmc $set --kernel-trace -d /tmp/p/sq$i -o sq -- python3 $R/bench.py --steps 4 --warmup 4 --no-cpu-baseline --no-ops-roofline --no-latency-sweep --throttle "$@" > $R/gpurun_out/${TAG}_sq$i.log 2>&1 && break
echo "sq pass $i attempt $attempt failed" >> $R/gpurun_out/${TAG}_sq_retries.log
done
DB=$(find /tmp/p/sq$i -name "*.db" | head -1)
[ -n "$DB" ] && python3 $R/tools/rocpd_sq.py $DB > $R/gpurun_out/${TAG}_sq$i.txt 2>&1
done
python3 $R/tools/rocpd_sq.py --merge $R/gpurun_out/${TAG}_sq1.txt $R/gpurun_out/${TAG}_sq2.txt $R/gpurun_out/${TAG}_sq3.txt $R/gpurun_out/${TAG}_sq4.txt > $R/gpurun_out/${TAG}_sq.txt 2>&1
