#!/bin/bash
out=gpurun_out/k9pmc
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VMEM" \
           "SQ_LDS_ADDR_CONFLICT SQ_INSTS_BRANCH SQ_WAVES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT TCC_MISS TCC_REQ TCC_WRITE TCC_EA_WRREQ"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/pass$i" -o p -- python3 tools/k9_probe.py "$@" > "$out/pass$i.log" 2> "$out/pass$i.err"
done
python3 tools/pmc_summary.py "$out" partition_scatter partition_hist > "$out/summary.txt" 2>&1
cat "$out/summary.txt"
