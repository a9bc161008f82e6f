#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 300 tests/cpp/bin/partition_operator_test 2>&1 | tail -30
timeout 900 python -m pytest tests/test_host_layer.py tests/test_abi.py tests/test_gpu_select.py tests/test_gpu_edge_cases.py -m gpu -x -q 2>&1 | tail -15
for cfg in "4 64" "4 256"; do
QSX_TEST_PROFILE=1 timeout 300 tests/cpp/bin/headline_operators_bench 1000000 100000000 600000000 5 2 $cfg 2>&1 | tail -9
done
