#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 300 tests/cpp/bin/hash_join_operator_test 2>&1 | tail -5
timeout 1200 python -m pytest tests/test_gpu_join.py tests/test_gpu_full_size.py tests/test_gpu_agg_jit.py tests/test_host_layer.py -m gpu -x -q 2>&1 | tail -8
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-operators 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value']/1e9, d['ms_per_step'], d['phases_ms']); print({k:(round(v['ms'],3) if isinstance(v,dict) else round(v,3)) for k,v in d['probe']['variants'].items()})"
done
QSX_JOIN_ADAPTIVE=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-operators 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('adaptive off', d['value']/1e9, d['ms_per_step'], d['phases_ms'])"
