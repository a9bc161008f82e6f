#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03m
mkdir -p $out
for cfg in c4 c5; do
  QSX_BENCH_FORCE_DISTRIBUTED=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --config $cfg > $out/bench_$cfg.json 2> $out/bench_$cfg.err; echo "rc=$?"; python3 -c "
import json,sys
d=json.loads([l for l in open('$out/bench_$cfg.json') if l.startswith('{')][-1]); print(d['ms_per_step'], d['value']/1e9, json.dumps(d['phases_ms'], indent=1))"
done
timeout 1200 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_two_ranks.py -m gpu -x -q 2>&1 | tail -8
