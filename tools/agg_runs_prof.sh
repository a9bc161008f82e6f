cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d gpurun_out/runsprof -- python3 tools/agg_runs_time.py > gpurun_out/runsprof.log 2>&1
python3 tools/rocpd_kernel_stats.py "$(find gpurun_out/runsprof -name '*.db' | head -1)" | grep -E "qsx|kernel " | head -8
rm -rf gpurun_out/runsprof
