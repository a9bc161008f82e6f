# kernel trace of the Q3 operator plan (tests/cpp/tpch_q3_plan_test.cpp) at 60 M lineitems in 120 K-row blocks: which kernels
# the work orders per block and per run of blocks launch, and how often
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/q3plan
rm -rf gpurun_out/q3plan/trace
rocprofv3 --kernel-trace --stats -d gpurun_out/q3plan/trace -- tests/cpp/bin/tpch_q3_plan_test 15000000 120000 > gpurun_out/q3plan/out.txt 2> gpurun_out/q3plan/err.txt
python3 tools/rocpd_kernel_stats.py "$(find gpurun_out/q3plan/trace -name '*.db' | head -1)" > gpurun_out/q3plan/kernel_stats.txt 2>&1
grep "Q3 plan" gpurun_out/q3plan/out.txt
head -45 gpurun_out/q3plan/kernel_stats.txt | cut -c1-150
rm -rf gpurun_out/q3plan/trace
