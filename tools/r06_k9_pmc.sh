#!/bin/bash
# round 6: SQ counters of the K9 kernels (tools/k9_probe.py), one rocprofv3 pass per counter set
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06_k9_pmc; mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VMEM" \
           "SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/pass$i" -o p -- python3 tools/k9_probe.py "$@" > "$out/pass$i.json" 2> "$out/pass$i.err"
done
python3 tools/pmc_summary.py "$out" partition_scatter partition_hist > "$out/summary.txt" 2>&1
cat "$out/summary.txt"
