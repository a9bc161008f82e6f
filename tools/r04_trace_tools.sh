#!/bin/bash
# kernel traces (rocprofv3 --kernel-trace --stats) of the per-topic tools whose kernels changed late in round 4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04t
mkdir -p $out
for t in agg_coded_probe agg_dense_probe agg_dir_probe; do
  rocprofv3 --kernel-trace --stats -d $out/trace_$t -- python3 tools/$t.py > $out/$t.json 2> $out/$t.err
  python3 tools/rocpd_kernel_stats.py "$(find $out/trace_$t -name '*.db' | head -1)" 2>/dev/null | grep -E "^kernel|qsx" | head -8 > $out/kernel_stats_$t.txt
  rm -rf $out/trace_$t
  cat $out/kernel_stats_$t.txt | cut -c 1-160
done
