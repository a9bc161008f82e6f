#!/bin/bash
# round 5: the failure-agreement paths on the GPU box (world 2 / 3 over the loopback transport), C++ rank processes included
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05b
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_comm.py tests/test_gpu_two_ranks.py tests/test_host_layer.py -x -q -m gpu --durations=15 > $out/pytest_comm.log 2>&1; tail -30 $out/pytest_comm.log
timeout 300 tests/cpp/bin/partitioned_ranks_test > $out/partitioned_ranks.log 2>&1; echo "partitioned_ranks_test rc=$?"; grep -a "failed round\|world\|FAIL\|failures" $out/partitioned_ranks.log | head -20
