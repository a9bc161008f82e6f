for v in 2 4; do for acc in 2 4 8 16; do
echo "V=$v ACC=$acc"; QSX_AGG_ROWS_PER_THREAD=$v QSX_AGG_ACC_KIB=$acc QSX_AGG_BLOCKS_PER_CU=8 python tools/agg_interp.py 2>&1 | grep -E "Q1 interpreter|Q1 plan"
done; done
