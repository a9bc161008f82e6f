#!/bin/bash
# gpurun_out/r06m (tools/r06_measure.sh) -> profiles/r06_*; profiles/traffic.json from the PMC summary.
cd "$(dirname "$0")/.." || exit 1
m=gpurun_out/r06m
cp $m/bench_headline.json profiles/r06_bench_line.json
cp $m/kernel_stats.txt profiles/r06_kernel_stats.txt
for c in c4 c5; do
  cp $m/kernel_stats_$c.txt profiles/r06_kernel_stats_$c.txt
  cp $m/kernel_stats_operators_$c.csv profiles/r06_kernel_stats_operators_$c.csv
  cp $m/bench_$c.json profiles/r06_bench_$c.json
  grep '^{' $m/operators_$c.json | tail -1 > profiles/r06_operators_$c.json
done
cp $m/pmc_summary.txt profiles/r06_pmc_summary.txt
for f in agg_large_groups agg_large_groups_one_pass agg_large_groups_two_levels_always join_coded_probe hashed_cycle k9_probe k9_blocks_probe lip_build_probe agg_filtered_groups agg_family_probe probe_hashed_sparse agg_coded_probe probe_small_tables agg_dir_probe agg_dense_probe bench_ops; do
  cp $m/$f.jsonl profiles/r06_$f.jsonl
done
cp $m/two_level_kernel_stats.csv profiles/r06_two_level_kernel_stats.csv
head -24 gpurun_out/r06m_console.log > profiles/r06_pytest_gpu.txt
python tools/update_traffic.py profiles/r06_pmc_summary.txt r06_pmc_summary.txt | cut -c1-200
python tools/design_table.py r06 | tail -1
