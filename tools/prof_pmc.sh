#!/bin/bash
# Runs the PMC passes for the two hot kernels (separate rocprofv3 runs per counter set, as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE cannot share a pass).
# usage (on the GPU box, from the repo root): tools/prof_pmc.sh <outdir> [extra bench args]
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT TCC_MISS TCC_REQ" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/pass$i" -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-probe-variants "$@" > "$out/pass$i.json" 2> "$out/pass$i.err"
done
python3 tools/pmc_summary.py "$out" agg_hash probe_kernel build_kernel radix > "$out/summary.txt" 2>&1
cat "$out/summary.txt"
