#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_join.py -m gpu -x -q 2>&1 | tail -15
timeout 600 python tools/probe_sliced.py 100000000 1000000 4000000 8000000 > gpurun_out/r03/probe_sliced.jsonl 2> gpurun_out/r03/probe_sliced.err; cat gpurun_out/r03/probe_sliced.jsonl; tail -3 gpurun_out/r03/probe_sliced.err
