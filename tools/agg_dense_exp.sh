#!/bin/bash
# What the parts of K7 (dense per-row path, clustered keys) cost: the run-time shape rebuilt (hipRTC) with parts compiled out.
# usage (GPU box, repo root): tools/agg_dense_exp.sh [rows_millions]
rows=${1:-200}
for opt in "" "-DQSX_EXP_NO_DENSE_ATOMICS" "-DQSX_EXP_NO_COMPUTE" "-DQSX_EXP_STAGE_ONCE" "-DQSX_EXP_STAGE_ONCE -DQSX_EXP_NO_DENSE_ATOMICS"; do
  echo "== QSX_JIT_OPTIONS=$opt"
  QSX_JIT_COMPILER=hiprtc QSX_JIT_OPTIONS="$opt" timeout -s KILL 120 python3 tools/agg_dense_probe.py $rows 2>/dev/null | tail -1
done
