#!/bin/bash
# last pass of round 4 on the GPU box (after the final kernel-source change): PMC passes -> profiles/traffic.json's inputs,
# kernel trace of the headline, the bench lines that quote them.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04f
mkdir -p $out
tools/prof_pmc.sh $out/pmc --no-operators > /dev/null 2>&1
python3 tools/pmc_summary.py $out/pmc agg_hash dense_probe probe_fp probe_kernel dense_build build_kernel gather_kernel cover_probe > $out/pmc_summary.txt 2>&1
find $out/pmc -name '*.csv' -size +1M -delete; find $out/pmc -name '*.db' -delete
python3 tools/update_traffic.py $out/pmc_summary.txt r04_pmc_summary.txt > /dev/null && cp profiles/traffic.json $out/traffic.json
rocprofv3 --kernel-trace --stats -d $out/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-operators > $out/bench_traced.json 2> $out/bench_traced.err
python3 tools/rocpd_kernel_stats.py "$(find $out/trace -name '*.db' | head -1)" > $out/kernel_stats.txt 2>&1; head -6 $out/kernel_stats.txt
rm -rf $out/trace
timeout 900 python bench.py --steps 25 --warmup 5 2> $out/bench_headline.err | tail -1 > $out/bench_headline.json; tail -c 300 $out/bench_headline.json
for tr in torch capi; do timeout 600 python bench.py --steps 10 --warmup 3 --config c4 --transport $tr 2>/dev/null | tail -1 > $out/bench_c4_$tr.json; done
timeout 300 python tools/probe_hashed_sparse.py > $out/probe_hashed_sparse.jsonl 2>/dev/null
timeout 400 python tools/bench_ops.py > $out/bench_ops.jsonl 2>/dev/null; wc -l $out/bench_ops.jsonl
