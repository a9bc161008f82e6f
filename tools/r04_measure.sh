#!/bin/bash
# round 4 measurement pass on the GPU box: full GPU suite, bench lines of the three configs (both transports for c4 / c5),
# kernel trace of the headline, PMC passes (aggregation + probe), the probe / aggregation micro-measurements, bench_ops.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04n
mkdir -p $out $out/pmc_sparse
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E " passed| failed|rror" | tail -5 > $out/pytest_gpu.log; cat $out/pytest_gpu.log
timeout 900 python bench.py --steps 25 --warmup 5 2> $out/bench_headline.err | tail -1 > $out/bench_headline.json; tail -c 300 $out/bench_headline.json; tail -2 $out/bench_headline.err
for cfg in c4 c5; do
  for tr in torch capi; do
    timeout 600 python bench.py --steps 10 --warmup 3 --config $cfg --transport $tr 2> $out/bench_${cfg}_$tr.err | tail -1 > $out/bench_${cfg}_$tr.json; tail -c 200 $out/bench_${cfg}_$tr.json; echo
  done
done
# two ranks on this one GPU (rehearsal: loopback transport, gloo control plane): the N > 1 path of every configuration, self-launched
for cfg in headline c4 c5; do
  QSX_BENCH_SHARED_GPU=1 QSX_ALLOW_TEST_TRANSPORT=1 QSX_RCCL_LIBRARY=$PWD/tests/cpp/bin/libloopback_rccl.so timeout 600 python bench.py --gpus 2 --transport capi --config $cfg --steps 3 --warmup 1 --no-cpu-baseline 2> $out/bench_${cfg}_2ranks_shared_gpu.err | tail -1 > $out/bench_${cfg}_2ranks_shared_gpu.json; tail -c 200 $out/bench_${cfg}_2ranks_shared_gpu.json; echo
done
QSX_BENCH_FORCE_DISTRIBUTED=1 timeout 600 python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --transport capi 2> $out/bench_headline_dist1_capi.err | tail -1 > $out/bench_headline_dist1_capi.json; tail -c 200 $out/bench_headline_dist1_capi.json; echo
rocprofv3 --kernel-trace --stats -d $out/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-operators > $out/bench_traced.json 2> $out/bench_traced.err
python3 tools/rocpd_kernel_stats.py "$(find $out/trace -name '*.db' | head -1)" > $out/kernel_stats.txt 2>&1; head -12 $out/kernel_stats.txt
rm -rf $out/trace
tools/prof_pmc.sh $out/pmc --no-operators > /dev/null 2>&1
python3 tools/pmc_summary.py $out/pmc agg_hash dense_probe probe_fp probe_kernel dense_build build_kernel gather_kernel cover_probe > $out/pmc_summary.txt 2>&1; grep -A 3 "agg_hash_shape_fixed" $out/pmc_summary.txt | head -8
find $out/pmc -name '*.csv' -size +1M -delete; find $out/pmc -name '*.db' -delete
# the hashed table over sparse keys under the counters (its launches are not part of the headline step)
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT TCC_MISS TCC_REQ"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/pmc_sparse/pass$i" -o p -- python3 tools/probe_hashed_sparse.py > "$out/pmc_sparse/pass$i.json" 2> "$out/pmc_sparse/pass$i.err"
done
python3 tools/pmc_summary.py $out/pmc_sparse probe_fp build_kernel > $out/pmc_summary_sparse_probe.txt 2>&1; head -12 $out/pmc_summary_sparse_probe.txt
find $out/pmc_sparse -name '*.csv' -size +1M -delete; find $out/pmc_sparse -name '*.db' -delete
QSX_TEST_PROFILE=1 tests/cpp/bin/headline_operators_bench 1000000 100000000 600000000 25 5 8 256 > $out/operators_profile.txt 2>&1; tail -9 $out/operators_profile.txt | cut -c 1-200
for t in probe_hashed_sparse agg_coded_probe agg_wide agg_dense_small probe_project agg_dir_probe; do timeout 300 python tools/$t.py > $out/$t.jsonl 2>/dev/null; done
timeout 400 python tools/bench_ops.py > $out/bench_ops.jsonl 2>/dev/null; wc -l $out/*.jsonl
