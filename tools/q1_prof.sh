# kernel trace of tools/q1_pipeline.py (of the tree given as $1, default this one): every aggregation / select launch in order
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tree=${1:-.}
mkdir -p gpurun_out/q1prof
rm -rf gpurun_out/q1prof/trace
rocprofv3 --kernel-trace --stats -d gpurun_out/q1prof/trace -- python3 $tree/tools/q1_pipeline.py 100 > gpurun_out/q1prof/out.json 2> gpurun_out/q1prof/err.txt
python3 tools/rocpd_kernel_list.py "$(find gpurun_out/q1prof/trace -name '*.db' | head -1)" qsx_jit_agg select_packed agg_hash
cut -c1-330 gpurun_out/q1prof/out.json
rm -rf gpurun_out/q1prof/trace
