# Sweep of the aggregation tuning knobs after the interpreter's configuration moved behind a pointer.
for rpt in 2 4; do for buf in 1 2; do for acc in 8 16; do for bpc in 3 4 6; do
  echo "rows_per_thread=$rpt buffers=$buf acc_kib=$acc blocks_per_cu=$bpc: $(QSX_AGG_ROWS_PER_THREAD=$rpt QSX_AGG_BUFFERS=$buf QSX_AGG_ACC_KIB=$acc QSX_AGG_BLOCKS_PER_CU=$bpc python tools/agg_interp.py 200000000 2>/dev/null | head -2 | awk '{print $1,$2,$3,$4}' | tr '\n' ';')"
done; done; done; done
