cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d gpurun_out/k9 -- python3 tools/k9_probe.py > gpurun_out/k9.log 2>&1
python3 tools/rocpd_kernel_stats.py "$(find gpurun_out/k9 -name '*.db' | head -1)" | grep -E "partition|scan|kernel " | head
tail -1 gpurun_out/k9.log
rm -rf gpurun_out/k9
