#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for cfg in "4 64" "4 256" "4 1024" "8 256"; do
QSX_TEST_PROFILE=1 timeout 300 tests/cpp/bin/headline_operators_bench 1000000 100000000 600000000 5 2 $cfg 2>&1 | tail -12
done
timeout 300 tests/cpp/bin/work_order_runs_test 2>&1 | tail -12
