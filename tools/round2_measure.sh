# round 2: every measurement DESIGN.md / profiles/README.md quote, in one GPU session
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r02
mkdir -p $o
python3 tools/bench_ops.py > $o/bench_ops.jsonl 2> $o/bench_ops.err
python3 tools/agg_midsize.py 200000000 20 50 70 100 > $o/agg_midsize.jsonl 2>&1
python3 tools/agg_midsize.py 200000000 70 100 --sparse >> $o/agg_midsize.jsonl 2>&1
python3 tools/probe_sliced.py > $o/probe_sliced.jsonl 2>&1
python3 tools/block_granularity.py > $o/block_granularity.jsonl 2>&1
python3 tools/q1_pipeline.py 100 > $o/q1_pipeline.jsonl 2>&1
python3 tools/q1_pipeline.py 100 int >> $o/q1_pipeline.jsonl 2>&1
python3 tools/q3_pipeline.py 100 nolip fused types > $o/q3_pipeline.jsonl 2>&1
python3 tools/q3_pipeline.py 100 nolip fused >> $o/q3_pipeline.jsonl 2>&1
python3 tools/q3_pipeline.py 100 lip fused types >> $o/q3_pipeline.jsonl 2>&1

python3 tools/agg_large_groups.py > $o/agg_large_groups.jsonl 2>&1
# the operator layer: work orders per block against work orders per run of blocks; allocation costs; the allocator finding
tests/cpp/bin/work_order_runs_test > $o/work_order_runs.txt 2>&1
tests/cpp/bin/tpch_types_operator_test 2>&1 | grep "work order" >> $o/work_order_runs.txt
tests/cpp/bin/tpch_q3_plan_test 1500000 120000 2>&1 | grep "Q3 plan" >> $o/work_order_runs.txt
tests/cpp/bin/tpch_q3_plan_test 15000000 120000 2>&1 | grep "Q3 plan" >> $o/work_order_runs.txt
for u in alloc_cost pool_readback; do [ -x tools/ubench/$u ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o tools/ubench/$u tools/ubench/$u.hip -lpthread; done
tools/ubench/alloc_cost > $o/alloc_cost.jsonl 2>&1
(for b in 8 65536 4194304; do timeout 300 tools/ubench/pool_readback 4 100000 $b 1 1 0 0; done
 echo "no plain hipMalloc/hipFree next to it:"; timeout 300 tools/ubench/pool_readback 4 100000 8 1 0 0 0
 echo "plain allocations instead of the pool:"; timeout 300 tools/ubench/pool_readback 4 100000 8 0 1 0 0
 echo "release threshold raised:"; timeout 300 tools/ubench/pool_readback 4 100000 8 1 1 0 1) > $o/pool_readback.txt 2>&1
grep -h "^{" $o/*.jsonl | cut -c1-200 | tail -30
