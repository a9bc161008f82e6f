#!/bin/bash
# round 5, first pass on the GPU box: smoke, the bench line with its `secondary` block (wall time), kernel trace of the same command
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05a
mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
t0=$(date +%s.%N); timeout 1200 python bench.py --steps 20 --warmup 5 2> $out/bench.err | tail -1 > $out/bench_line.json; echo "bench wall $(echo "$(date +%s.%N) - $t0" | bc) s"
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r05a/bench_line.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"], "frac", d["roofline"]["frac"])
for k,v in d.get("secondary",{}).items():
    if isinstance(v,dict):
        print(k, {x: v.get(x) for x in ("ms","checked","wall_s","error")}, "frac", (v.get("roofline") or {}).get("frac"), "cpu", (v.get("cpu_baseline") or {}).get("value"))
    else:
        print(k, v)
PY
rocprofv3 --kernel-trace --stats -d $out/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-operators --no-probe-variants > $out/bench_traced.json 2> $out/bench_traced.err
python3 tools/rocpd_kernel_stats.py "$(find $out/trace -name '*.db' | head -1)" > $out/kernel_stats.txt 2>&1; head -40 $out/kernel_stats.txt
rm -rf $out/trace
