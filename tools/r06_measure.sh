#!/bin/bash
# round 6 measurement pass on the GPU box: the full GPU suite (durations), the bench line with its `secondary` block, kernel
# traces (headline + secondary; c4; c5; the operators' C4 / C5), PMC passes for the headline (-> profiles/traffic.json), the
# per-topic tools of the round (two-level aggregation, joins on code stripes, hashed build cycle, K9, the AOT family) and
# tools/bench_ops.py.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06m
rm -rf $out; mkdir -p $out
t0=$(date +%s)
timeout 1800 python -m pytest tests -m gpu -x -q --durations=20 > $out/pytest_gpu_full.log 2>&1
grep -E " passed| failed|rror" $out/pytest_gpu_full.log | tail -3; echo "gpu suite wall: $(( $(date +%s) - t0 )) s"
grep -A 22 "slowest 20" $out/pytest_gpu_full.log | cut -c1-150
t0=$(date +%s)
timeout 900 python bench.py --steps 20 --warmup 5 2> $out/bench_headline.err | tail -1 > $out/bench_headline.json; echo "bench wall: $(( $(date +%s) - t0 )) s"; tail -2 $out/bench_headline.err
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r06m/bench_headline.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"], "frac", d["roofline"]["frac"], "probe", d["probe"]["ms"], d["probe"]["roofline"]["frac"])
for k,v in d.get("secondary",{}).items():
    if isinstance(v,dict):
        print(k, {x: v.get(x) for x in ("ms","checked","wall_s","error")}, "frac", (v.get("roofline") or {}).get("frac"), "cpu", (v.get("cpu_baseline") or {}).get("value"))
        for kk, vv in v.items():
            if isinstance(vv, dict) and ("ms_per_step" in vv or "ms" in vv) and kk not in ("roofline", "cpu_baseline"):
                print("    ", kk, {x: vv.get(x) for x in ("ms", "ms_per_step", "checked", "error")})
    else:
        print(k, v)
PY
rocprofv3 --kernel-trace --stats -d $out/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-operators > $out/bench_traced.json 2> $out/bench_traced.err
python3 tools/rocpd_kernel_stats.py "$(find $out/trace -name '*.db' | head -1)" > $out/kernel_stats.txt 2>&1; grep -v "_ZN2at\|rocclr\|rocprim\|cuda_kernel" $out/kernel_stats.txt | head -30 | cut -c1-160
rm -rf $out/trace
for cfg in c4 c5; do
  timeout 600 python bench.py --steps 10 --warmup 3 --config $cfg 2> $out/bench_$cfg.err | tail -1 > $out/bench_$cfg.json; tail -c 150 $out/bench_$cfg.json; echo
  rocprofv3 --kernel-trace --stats -d $out/trace_$cfg -- python3 bench.py --config $cfg --steps 5 --warmup 2 > /dev/null 2> $out/trace_$cfg.err
  python3 tools/rocpd_kernel_stats.py "$(find $out/trace_$cfg -name '*.db' | head -1)" > $out/kernel_stats_$cfg.txt 2>&1
  rm -rf $out/trace_$cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_ops_$cfg -o ops -- tests/cpp/bin/partitioned_operators_bench $cfg $([ $cfg = c4 ] && echo 18750000 || echo 37.5) 5 2 4 1024 > $out/operators_$cfg.json 2> $out/operators_$cfg.err
  cp "$(find $out/trace_ops_$cfg -name '*kernel_stats.csv' | head -1)" $out/kernel_stats_operators_$cfg.csv 2>/dev/null
  rm -rf $out/trace_ops_$cfg
  tail -c 300 $out/operators_$cfg.json; echo
done
tools/prof_pmc.sh $out/pmc --no-operators --no-secondary > /dev/null 2>&1
python3 tools/pmc_summary.py $out/pmc agg_hash dense_probe probe_fp dense_build build_kernel > $out/pmc_summary.txt 2>&1; grep -A 3 "agg_hash_shape_fixed" $out/pmc_summary.txt | head -8
find $out/pmc -name '*.csv' -size +1M -delete; find $out/pmc -name '*.db' -delete
# the round's topics
timeout 600 python tools/agg_large_groups.py > $out/agg_large_groups.jsonl 2>/dev/null
QSX_AGG_TWO_LEVEL_MIN_GROUPS=0 timeout 600 python tools/agg_large_groups.py > $out/agg_large_groups_one_pass.jsonl 2>/dev/null
QSX_AGG_TWO_LEVEL_MIN_GROUPS=50000 QSX_AGG_TWO_LEVEL_SAMPLE=0 timeout 600 python tools/agg_large_groups.py > $out/agg_large_groups_two_levels_always.jsonl 2>/dev/null
for f in agg_large_groups agg_large_groups_one_pass agg_large_groups_two_levels_always; do echo "== $f"; grep GENERIC $out/$f.jsonl | cut -c18-60,95-; done
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/trace_two_level" -o two_level -- python3 "$GRAFT_REPO_ROOT/tools/agg_large_groups.py" random-generic > /dev/null 2>&1)
cp "$(find $out/trace_two_level -name '*kernel_stats.csv' | head -1)" $out/two_level_kernel_stats.csv; rm -rf $out/trace_two_level; head -9 $out/two_level_kernel_stats.csv | cut -c1-170
timeout 600 python tools/join_coded_probe.py > $out/join_coded_probe.jsonl 2>/dev/null; cat $out/join_coded_probe.jsonl
for t in hashed_cycle k9_probe k9_blocks_probe lip_build_probe agg_family_probe probe_hashed_sparse agg_coded_probe probe_small_tables agg_dir_probe agg_dense_probe; do timeout 300 python tools/$t.py > $out/$t.jsonl 2>/dev/null; done
timeout 300 python tools/agg_filtered_groups.py > $out/agg_filtered_groups.jsonl 2>/dev/null; QSX_AGG_FILTER_COMPACT=0 timeout 300 python tools/agg_filtered_groups.py >> $out/agg_filtered_groups.jsonl 2>/dev/null
timeout 600 python tools/bench_ops.py > $out/bench_ops.jsonl 2>/dev/null; wc -l $out/*.jsonl | tail -20
grep "1,000,000\|10,000,000" $out/bench_ops.jsonl | cut -c1-260
