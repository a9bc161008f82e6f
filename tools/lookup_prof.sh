# Kernel trace of tools/lookup_probe.py: per-kernel durations of the lookup kernels.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/lookupprof
rocprofv3 --kernel-trace --stats -d gpurun_out/lookupprof/trace -- python3 tools/lookup_probe.py ${1:-600} > gpurun_out/lookupprof/out.json 2> gpurun_out/lookupprof/err.txt
python3 tools/rocpd_kernel_stats.py "$(find gpurun_out/lookupprof/trace -name '*.db' | head -1)" > gpurun_out/lookupprof/kernel_stats.txt 2>&1
grep "qsx" gpurun_out/lookupprof/kernel_stats.txt | head -30
rm -rf gpurun_out/lookupprof/trace
