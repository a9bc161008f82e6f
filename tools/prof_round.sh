cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof_r02
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r02/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_r02/bench.json 2> gpurun_out/prof_r02/bench.err
python3 tools/rocpd_kernel_stats.py "$(find gpurun_out/prof_r02/trace -name '*.db' | head -1)" > gpurun_out/prof_r02/kernel_stats.txt 2>&1
cat gpurun_out/prof_r02/kernel_stats.txt | head -30
if [ "$1" != "--no-pmc" ]; then
tools/prof_pmc.sh gpurun_out/prof_r02/pmc > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/prof_r02/pmc agg_hash dense_probe dense_build > gpurun_out/prof_r02/pmc_summary.txt 2>&1
head -80 gpurun_out/prof_r02/pmc_summary.txt
fi
rm -rf gpurun_out/prof_r02/trace
