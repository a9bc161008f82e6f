#!/bin/bash
# Records the run-time plan shapes the GPU suite, the bench configurations and the measurement tools ask for (on the GPU box:
# the launch geometry folded into a shape is decided with the device at hand) into gpurun_out/jit_shapes/; copy them to
# quickstep_amd/csrc/jit_shapes/ and the build compiles them into quickstep_amd/lib/jit_cache (__graft_entry__.warm_jit_cache).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=$PWD/gpurun_out/jit_shapes
mkdir -p $out
export QSX_JIT_RECORD_DIR=$out
timeout 1500 python -m pytest tests -q -m gpu -x --durations=30 > gpurun_out/record_pytest.log 2>&1; tail -40 gpurun_out/record_pytest.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-operators --no-probe-variants > gpurun_out/record_bench.log 2>&1; tail -c 300 gpurun_out/record_bench.log
for t in agg_coded_probe.py agg_dir_probe.py agg_dense_probe.py agg_dense_small.py agg_wide.py bench_ops.py q3_pipeline.py; do
  [ -f tools/$t ] && timeout 300 python tools/$t > /dev/null 2>&1
done
ls $out | wc -l
