#!/bin/bash
# PMC passes over the Q3 pipeline (tools/q3_pipeline.py, SF $2 default 30): what bounds the lookup kernels (LIP probe, dense probe).
out=$1; SF=${2:-30}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum" \
           "FETCH_SIZE TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "WRITE_SIZE"; do   # FETCH_SIZE and WRITE_SIZE cannot share a pass (tools/prof_pmc.sh)
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/pass$i" -o p -- python3 tools/q3_pipeline.py $SF > "$out/pass$i.log" 2> "$out/pass$i.err"
done
python3 tools/pmc_summary.py "$out" lip_probe_kernel dense_probe_kernel select_packed > "$out/summary.txt" 2>&1
cat "$out/summary.txt"
rm -rf "$out"/pass*/
