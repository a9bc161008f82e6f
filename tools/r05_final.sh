#!/bin/bash
# the round's closing pass: GPU suite, the bench line, the kernel trace of the coded-blocks probe
cd $GRAFT_REPO_ROOT
out=gpurun_out/r05_final
rm -rf $out; mkdir -p $out
d0=$(date +%s)
timeout 1500 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1
echo "pytest rc $? wall $(( $(date +%s) - d0 )) s" >> $out/pytest_gpu.log
tail -3 $out/pytest_gpu.log
d0=$(date +%s)
python bench.py --steps 20 --warmup 5 > $out/bench_line.json 2> $out/bench_line.err
echo "bench rc $? wall $(( $(date +%s) - d0 )) s"
bash tools/r05_blocks_trace.sh > $out/blocks_trace.txt 2>&1
cp gpurun_out/prof_blocks/run_kernel_stats.csv $out/blocks_kernel_stats.csv
tail -4 $out/blocks_trace.txt | cut -c1-300
