# Sweep of the accumulator LDS budget / workgroups per CU for the AOT Q1 kernel (200 M rows).
for acc in 2 4 5 8; do for bpc in 4 5; do
  echo "acc_kib=$acc blocks_per_cu=$bpc: $(QSX_AGG_ACC_KIB=$acc QSX_AGG_BLOCKS_PER_CU=$bpc python tools/agg_interp.py 200000000 2>/dev/null | head -1)"
done; done
