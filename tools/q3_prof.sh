# Kernel trace of the Q3 pipeline (tools/q3_pipeline.py) at SF $1 (default 100): per-kernel durations.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
SF=${1:-100}
mkdir -p gpurun_out/q3prof
rocprofv3 --kernel-trace --stats -d gpurun_out/q3prof/trace -- python3 tools/q3_pipeline.py $SF > gpurun_out/q3prof/q3.json 2> gpurun_out/q3prof/q3.err
python3 tools/rocpd_kernel_stats.py "$(find gpurun_out/q3prof/trace -name '*.db' | head -1)" > gpurun_out/q3prof/kernel_stats.txt 2>&1
head -50 gpurun_out/q3prof/kernel_stats.txt
rm -rf gpurun_out/q3prof/trace
