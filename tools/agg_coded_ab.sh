#!/bin/bash
# A/B of the coded Q1 aggregation (tools/agg_coded_probe.py) under the launch switches: register groups, tile buffers
# (0 = a second buffer when three workgroups per CU still fit), rows per thread, accumulator replication.
# usage (GPU box, repo root): tools/agg_coded_ab.sh [rows_millions]
rows=${1:-600}
for env in "QSX_AGG_REG_GROUPS=0" "QSX_AGG_REG_GROUPS=1" "QSX_AGG_BUFFERS=0" "QSX_AGG_JIT_ROWS=2 QSX_AGG_BLOCKS_PER_CU=8" "QSX_AGG_ACC_KIB=8"; do
  echo "== $env"
  env $env QSX_DEBUG_LAUNCH=1 timeout -s KILL 120 python3 tools/agg_coded_probe.py $rows 2> /tmp/coded_ab.err | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print({k:(round(v,3) if isinstance(v,float) else v) for k,v in d.items() if 'ms' in k or 'same' in k})"
  grep "jit launch\|shape launch" /tmp/coded_ab.err | sort | uniq -c | head -4
done
