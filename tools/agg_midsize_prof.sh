cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d gpurun_out/mid -- python3 tools/agg_midsize_probe.py > gpurun_out/mid.log 2>&1
python3 tools/rocpd_kernel_stats.py "$(find gpurun_out/mid -name '*.db' | head -1)" | grep -E "qsx|kernel " | head -12
grep update gpurun_out/mid.log
rm -rf gpurun_out/mid
