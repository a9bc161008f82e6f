#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_comm.py tests/test_abi.py tests/test_host_layer.py tests/test_gpu_join.py -m gpu -x -q 2>&1 | tail -15
for cfg in "4 64" "4 256" "8 256"; do
QSX_TEST_PROFILE=1 timeout 300 tests/cpp/bin/headline_operators_bench 1000000 100000000 600000000 8 4 $cfg 2>&1 | tail -9
done
