import os, sys, torch
sys.path.insert(0, "/root/repo")
os.environ["QSX_AGG_JIT_MIN_ROWS"] = "0"
import quickstep_amd.capi as capi
from quickstep_amd import types as T
import bench
dev = torch.device("cuda", 0)
n = 200_000_000
cols = bench.gen_q1_columns_gpu(n, dev, 4)
g = torch.Generator(device=dev); g.manual_seed(1)
ship = torch.randint(19920101, 19981201, (n,), device=dev, generator=g, dtype=torch.int32)
def timed(fn, reps=3):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
base = dict(keys=[0, 1], instrs=[(T.EX_SUB, 0, T.const(0), T.col(4)), (T.EX_MUL, 1, T.col(3), T.temp(0)), (T.EX_ADD, 2, T.const(0), T.col(5)), (T.EX_MUL, 3, T.temp(1), T.temp(2))], consts=[1.0],
            aggs=[(T.AGG_SUM, T.col(2)), (T.AGG_SUM, T.col(3)), (T.AGG_SUM, T.temp(1)), (T.AGG_SUM, T.temp(3)), (T.AGG_AVG, T.col(2)), (T.AGG_AVG, T.col(3)), (T.AGG_AVG, T.col(4)), (T.AGG_COUNT_STAR, None)], est_groups=6)
c6 = [(T.CHAR, 1), (T.CHAR, 1)] + [(T.DOUBLE, None)] * 4
for name, layout, pred, data in (
    ("no predicate (6 columns)", c6, [], cols),
    ("predicate on qty (always true)", c6, [(2, T.LT, 1000.0)], cols),
    ("7th column staged, no predicate", c6 + [(T.INT, None)], [], cols + [ship]),
    ("predicate on the 7th column (98 %)", c6 + [(T.INT, None)], [(6, T.LE, 19980902)], cols + [ship]),
):
    os.environ["QSX_AGG_NO_SPECIALIZE"] = "1"
    st = capi.AggState(T.make_agg_config(T.AGG_COMPACT_KEY, layout, pred=pred, **base))
    os.environ.pop("QSX_AGG_NO_SPECIALIZE")
    print("%-42s %.3f ms" % (name, timed(lambda: st.update(data, n))), flush=True)
