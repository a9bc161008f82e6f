#!/bin/bash
# round 3 measurement pass on the GPU box: full GPU suite, bench lines of the three configs, kernel trace, PMC passes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03m
mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; tail -5 $out/pytest_gpu.log
timeout 900 python bench.py --steps 25 --warmup 5 > $out/bench_headline.json 2> $out/bench_headline.err; tail -c 600 $out/bench_headline.json; tail -3 $out/bench_headline.err
for cfg in c4 c5; do
  QSX_BENCH_FORCE_DISTRIBUTED=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --config $cfg > $out/bench_$cfg.json 2> $out/bench_$cfg.err; tail -c 400 $out/bench_$cfg.json
done
QSX_BENCH_FORCE_DISTRIBUTED=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_headline_dist1.json 2> $out/bench_headline_dist1.err; tail -c 300 $out/bench_headline_dist1.json
rocprofv3 --kernel-trace --stats -d $out/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-operators > $out/bench_traced.json 2> $out/bench_traced.err
python3 tools/rocpd_kernel_stats.py "$(find $out/trace -name '*.db' | head -1)" > $out/kernel_stats.txt 2>&1; head -25 $out/kernel_stats.txt
rm -rf $out/trace
tools/prof_pmc.sh $out/pmc --no-operators > /dev/null 2>&1
python3 tools/pmc_summary.py $out/pmc agg_hash dense_probe probe_kernel dense_build build_kernel gather_kernel > $out/pmc_summary.txt 2>&1; head -120 $out/pmc_summary.txt
find $out/pmc -name '*.csv' -size +1M -delete; find $out/pmc -name '*.db' -delete
