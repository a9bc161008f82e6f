#!/bin/bash
# Round 6: the two-level partitioned aggregation — its tests, the large-groups tool with and without it, and its kernels' times.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/two_level
timeout 900 python -m pytest tests/test_gpu_agg.py -x -q -k "two_level" > gpurun_out/two_level/tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/two_level/tests.log
if [ "$1" != "quick" ]; then
timeout 600 python tools/agg_large_groups.py > gpurun_out/two_level/large_groups.jsonl 2> gpurun_out/two_level/large_groups.err
QSX_AGG_TWO_LEVEL_MIN_GROUPS=0 timeout 600 python tools/agg_large_groups.py > gpurun_out/two_level/large_groups_one_pass.jsonl 2>/dev/null
QSX_AGG_TWO_LEVEL_MIN_GROUPS=50000 QSX_AGG_TWO_LEVEL_SAMPLE=0 timeout 600 python tools/agg_large_groups.py > gpurun_out/two_level/large_groups_always.jsonl 2>/dev/null
for f in large_groups large_groups_one_pass large_groups_always; do echo "== $f"; grep GENERIC gpurun_out/two_level/$f.jsonl | cut -c18-60,95-; done
fi
rm -rf gpurun_out/two_level/prof; cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/two_level/prof" -o two_level -- python3 "$GRAFT_REPO_ROOT/tools/agg_large_groups.py" random-generic > "$GRAFT_REPO_ROOT/gpurun_out/two_level/large_groups_prof.jsonl" 2>&1
cd "$GRAFT_REPO_ROOT"; f=$(find gpurun_out/two_level/prof -name '*kernel_stats.csv' | head -1); head -25 "$f" | cut -c1-220
