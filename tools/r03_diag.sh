#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03m
mkdir -p $out
for cfg in c4 c5; do
  QSX_BENCH_FORCE_DISTRIBUTED=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --config $cfg > $out/bench_$cfg.json 2> $out/bench_$cfg.err; echo "rc=$?"; tail -c 600 $out/bench_$cfg.json; tail -5 $out/bench_$cfg.err
done
QSX_BENCH_FORCE_DISTRIBUTED=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_headline_dist1.json 2> $out/bench_headline_dist1.err; echo "rc=$?"; tail -c 300 $out/bench_headline_dist1.json; tail -5 $out/bench_headline_dist1.err
