#!/bin/bash
# What the parts of the coded Q1 aggregation cost: the run-time shape rebuilt (hipRTC) with parts compiled out through the
# QSX_EXP_* hooks of csrc/agg_hash_update.hpp.  Results of the cut-down kernels are wrong by construction; only their time counts.
# (the plain 34 B/row leg runs as a run-time shape here too, so the switches reach it)
# usage (GPU box, repo root): tools/agg_coded_exp.sh [rows_millions]
rows=${1:-600}
for opt in "" "-DQSX_EXP_NO_LDS_ADD" "-DQSX_EXP_NO_COMPUTE" "-DQSX_EXP_STAGE_ONCE" "-DQSX_EXP_STAGE_ONCE -DQSX_EXP_NO_LDS_ADD"; do
  echo "== QSX_JIT_OPTIONS=$opt"
  QSX_PROBE_PLAIN_RUNTIME_SHAPE=1 QSX_JIT_COMPILER=hiprtc QSX_JIT_OPTIONS="$opt" QSX_AGG_REG_GROUPS=0 timeout -s KILL 120 python3 tools/agg_coded_probe.py $rows 2> /tmp/coded_exp.err | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print({k:(round(v,3) if isinstance(v,float) else v) for k,v in d.items() if 'ms' in k or 'same' in k})"
done
