#!/bin/bash
# round 3, first GPU pass: 2-rank tests, bench legs of every config on one GPU
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
timeout 600 python -m pytest tests/test_gpu_two_ranks.py tests/test_abi.py -m gpu -x -q 2>&1 | tail -15
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r03/bench_headline.json 2> gpurun_out/r03/bench_headline.err; tail -c 3000 gpurun_out/r03/bench_headline.json; tail -5 gpurun_out/r03/bench_headline.err
for cfg in c4 c5; do
  QSX_BENCH_FORCE_DISTRIBUTED=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 3 --warmup 1 --config $cfg > gpurun_out/r03/bench_$cfg.json 2> gpurun_out/r03/bench_$cfg.err; tail -c 2500 gpurun_out/r03/bench_$cfg.json; tail -5 gpurun_out/r03/bench_$cfg.err
done
QSX_BENCH_FORCE_DISTRIBUTED=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 3 --warmup 1 > gpurun_out/r03/bench_headline_dist1.json 2> gpurun_out/r03/bench_headline_dist1.err; tail -c 2000 gpurun_out/r03/bench_headline_dist1.json; tail -5 gpurun_out/r03/bench_headline_dist1.err
