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
grep -h "^{" $o/*.jsonl | cut -c1-200 | tail -30
