#!/bin/bash
# What the parts of the group-directory kernels cost (tools/agg_dir_probe.py with parts compiled out; with two tile buffers
# QSX_EXP_STAGE_ONCE leaves the second buffer unstaged: times only).
# usage (GPU box, repo root): tools/agg_dir_exp.sh [rows_millions]
rows=${1:-200}
for opt in "" "-DQSX_EXP_NO_COMPUTE" "-DQSX_EXP_STAGE_ONCE"; do
  echo "== QSX_JIT_OPTIONS=$opt"
  QSX_JIT_COMPILER=hiprtc QSX_JIT_OPTIONS="$opt" QSX_DEBUG_LAUNCH=1 timeout -s KILL 120 python3 tools/agg_dir_probe.py $rows 2>/tmp/dir_exp.err | tail -1 | cut -c 1-330
  grep "launch" /tmp/dir_exp.err | sort | uniq -c | sort -rn | head -3
done
