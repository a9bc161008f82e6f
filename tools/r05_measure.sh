#!/bin/bash
# round 5 measurement pass on the GPU box: the full GPU suite (with its durations and the plan shapes it asks for), the bench
# line with its `secondary` block, kernel traces (headline + secondary; c4; c5), PMC passes (headline; Q1 over code stripes;
# the LDS-resident probe), the per-topic tools.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05m
mkdir -p $out $out/jit_shapes $out/pmc_coded $out/pmc_lds_probe $out/pmc_sparse
export QSX_JIT_RECORD_DIR=$PWD/$out/jit_shapes
t0=$(date +%s)
timeout 1800 python -m pytest tests -m gpu -x -q --durations=25 > $out/pytest_gpu_full.log 2>&1
grep -E " passed| failed|rror" $out/pytest_gpu_full.log | tail -3; echo "gpu suite wall: $(( $(date +%s) - t0 )) s"
grep -A 27 "slowest 25" $out/pytest_gpu_full.log | cut -c1-150
t0=$(date +%s)
timeout 900 python bench.py --steps 20 --warmup 5 2> $out/bench_headline.err | tail -1 > $out/bench_headline.json; echo "bench wall: $(( $(date +%s) - t0 )) s"; tail -2 $out/bench_headline.err
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r05m/bench_headline.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"], "frac", d["roofline"]["frac"], "probe", d["probe"]["ms"], d["probe"]["roofline"]["frac"])
for k,v in d.get("secondary",{}).items():
    if isinstance(v,dict):
        print(k, {x: v.get(x) for x in ("ms","checked","wall_s","error")}, "frac", (v.get("roofline") or {}).get("frac"), "cpu", (v.get("cpu_baseline") or {}).get("value"))
    else:
        print(k, v)
PY
unset QSX_JIT_RECORD_DIR
rocprofv3 --kernel-trace --stats -d $out/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-operators > $out/bench_traced.json 2> $out/bench_traced.err
python3 tools/rocpd_kernel_stats.py "$(find $out/trace -name '*.db' | head -1)" > $out/kernel_stats.txt 2>&1; grep -v "_ZN2at\|rocclr\|rocprim\|cuda_kernel" $out/kernel_stats.txt | head -30 | cut -c1-160
rm -rf $out/trace
for cfg in c4 c5; do
  timeout 600 python bench.py --steps 10 --warmup 3 --config $cfg 2> $out/bench_$cfg.err | tail -1 > $out/bench_$cfg.json; tail -c 150 $out/bench_$cfg.json; echo
  rocprofv3 --kernel-trace --stats -d $out/trace_$cfg -- python3 bench.py --config $cfg --steps 5 --warmup 2 > /dev/null 2> $out/trace_$cfg.err
  python3 tools/rocpd_kernel_stats.py "$(find $out/trace_$cfg -name '*.db' | head -1)" > $out/kernel_stats_$cfg.txt 2>&1
  rm -rf $out/trace_$cfg
done
tools/prof_pmc.sh $out/pmc --no-operators --no-secondary > /dev/null 2>&1
python3 tools/pmc_summary.py $out/pmc agg_hash dense_probe probe_fp dense_build build_kernel > $out/pmc_summary.txt 2>&1; grep -A 3 "agg_hash_shape_fixed" $out/pmc_summary.txt | head -8
find $out/pmc -name '*.csv' -size +1M -delete; find $out/pmc -name '*.db' -delete
# the secondary's own kernels under the counters: Q1 over code stripes (agg_factored_direct_kernel) and the LDS-resident probe
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/pmc_coded/pass$i" -o p -- python3 tools/agg_coded_probe.py > "$out/pmc_coded/pass$i.json" 2> "$out/pmc_coded/pass$i.err"
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/pmc_lds_probe/pass$i" -o p -- python3 tools/probe_small_tables.py 100 > "$out/pmc_lds_probe/pass$i.json" 2> "$out/pmc_lds_probe/pass$i.err"
done
# the bucketed table over sparse keys behind its compact plane (probe_fp_kernel<..., kCompact>)
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT TCC_MISS TCC_REQ"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/pmc_sparse/pass$i" -o p -- python3 tools/probe_hashed_sparse.py > "$out/pmc_sparse/pass$i.json" 2> "$out/pmc_sparse/pass$i.err"
done
python3 tools/pmc_summary.py $out/pmc_sparse probe_fp build_kernel compact_build > $out/pmc_summary_sparse_probe.txt 2>&1; head -12 $out/pmc_summary_sparse_probe.txt
find $out/pmc_sparse -name '*.csv' -size +1M -delete; find $out/pmc_sparse -name '*.db' -delete
bash tools/r05_blocks_trace.sh > $out/blocks_trace.txt 2>&1; cp gpurun_out/prof_blocks/run_kernel_stats.csv $out/blocks_kernel_stats.csv; grep -E "coef|factored_direct" $out/blocks_trace.txt | cut -c1-200
python3 tools/pmc_summary.py $out/pmc_coded agg_factored qsx_jit_agg agg_hash > $out/pmc_summary_coded.txt 2>&1; head -14 $out/pmc_summary_coded.txt
python3 tools/pmc_summary.py $out/pmc_lds_probe lds_dense lds_bucket dense_probe probe_fp > $out/pmc_summary_lds_probe.txt 2>&1; head -14 $out/pmc_summary_lds_probe.txt
find $out/pmc_coded $out/pmc_lds_probe -name '*.csv' -size +1M -delete; find $out/pmc_coded $out/pmc_lds_probe -name '*.db' -delete
for t in probe_hashed_sparse agg_coded_probe probe_small_tables agg_dir_probe agg_dense_probe select_char_probe q1_predicate_probe; do timeout 300 python tools/$t.py > $out/$t.jsonl 2>/dev/null; done
timeout 400 python tools/bench_ops.py > $out/bench_ops.jsonl 2>/dev/null; wc -l $out/*.jsonl
ls $out/jit_shapes | wc -l
