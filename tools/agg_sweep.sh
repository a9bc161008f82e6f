for v in 2 4; do for b in 1 2; do for a in 6 12 24; do for m in 3 4 6; do
echo "V=$v BUF=$b ACC=$a MAXB=$m: $(QSX_AGG_ROWS_PER_THREAD=$v QSX_AGG_BUFFERS=$b QSX_AGG_ACC_KIB=$a QSX_AGG_BLOCKS_PER_CU=$m python tools/agg_probe.py 2>&1 | grep 'E full' )"
done; done; done; done
