#!/bin/bash
# kernel trace of tools/agg_coded_probe.py (Q1 over code stripes: one stripe, and a run of 4 MiB blocks with their own dictionaries)
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_blocks
rm -rf $out; mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 tools/agg_coded_probe.py > $out/probe.json 2> $out/probe.err
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(r["Name"][:120], r["Calls"], r["AverageNs"], r["MinNs"])
PY
cat $out/probe.json | tail -1 | cut -c1-400
