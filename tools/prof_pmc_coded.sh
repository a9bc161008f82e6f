#!/bin/bash
# PMC passes over the coded Q1 aggregation (tools/agg_coded_probe.py), one rocprofv3 run per counter set.
# usage (GPU box, repo root): tools/prof_pmc_coded.sh <outdir> [rows_millions]   (environment switches are inherited)
out=$1; rows=${2:-300}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VMEM" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INST_CYCLES_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/pass$i" -o p -- python3 tools/agg_coded_probe.py $rows > "$out/pass$i.json" 2> "$out/pass$i.err"
done
python3 tools/pmc_summary.py "$out" qsx_jit_agg agg_hash_shape > "$out/summary.txt" 2>&1
cat "$out/summary.txt"
