#!/bin/bash
# round 6: the hashed table's clear + build + first probe, phase by phase and kernel by kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06_cycle
mkdir -p $out
python3 tools/hashed_cycle.py 1000000 100000000 | tee $out/cycle.txt
python3 tools/hashed_cycle.py 1000000 100000000 sparse | tee -a $out/cycle.txt
rocprofv3 --kernel-trace --stats -d $out/trace -- python3 tools/hashed_cycle.py 1000000 100000000 > $out/traced.txt 2>&1
python3 tools/rocpd_kernel_stats.py "$(find $out/trace -name '*.db' | head -1)" > $out/kernel_stats.txt 2>&1; head -30 $out/kernel_stats.txt
rm -rf $out/trace
