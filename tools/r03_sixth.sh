#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
timeout 1200 python -m pytest tests/test_gpu_join.py tests/test_gpu_full_size.py tests/test_host_layer.py tests/test_gpu_q3_pipeline.py tests/test_gpu_partitioned_join.py -m gpu -x -q 2>&1 | tail -12
timeout 600 python tools/probe_sliced.py 100000000 1000000 > gpurun_out/r03/probe_tables.jsonl 2> gpurun_out/r03/probe_tables.err; cat gpurun_out/r03/probe_tables.jsonl; tail -3 gpurun_out/r03/probe_tables.err
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-operators 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value']/1e9, d['ms_per_step'], d['phases_ms']); print({k:(round(v['ms'],3) if isinstance(v,dict) else round(v,3)) for k,v in d['probe']['variants'].items()})"
