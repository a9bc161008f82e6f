#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
timeout 1200 python -m pytest tests/test_gpu_join.py tests/test_abi.py tests/test_gpu_two_ranks.py tests/test_gpu_partitioned_join.py -m gpu -x -q 2>&1 | tail -15
timeout 300 tests/cpp/bin/headline_operators_bench 1000000 100000000 600000000 5 2 4 64 2>&1 | tail -5
timeout 300 tests/cpp/bin/headline_operators_bench 1000000 100000000 600000000 5 2 8 64 2>&1 | tail -3
timeout 300 tests/cpp/bin/headline_operators_bench 1000000 100000000 600000000 5 2 4 256 2>&1 | tail -3
timeout 600 python tools/probe_sliced.py 100000000 1000000 > gpurun_out/r03/probe_tables.jsonl 2> gpurun_out/r03/probe_tables.err; cat gpurun_out/r03/probe_tables.jsonl; tail -3 gpurun_out/r03/probe_tables.err
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/r03/bench_headline.json 2> gpurun_out/r03/bench_headline.err; tail -c 4500 gpurun_out/r03/bench_headline.json; tail -5 gpurun_out/r03/bench_headline.err
