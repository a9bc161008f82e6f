#!/usr/bin/env python3
"""Times the aggregation interpreter (run-time configuration, QSX_AGG_NO_SPECIALIZE=1) against the AOT plan shape
on the Q1 shape and on a generic two-key configuration.  Tuning comes from QSX_AGG_* (see csrc/aggregate.hip)."""
import os
import sys

import torch

os.environ.setdefault("QSX_AGG_JIT_SYNC", "1")   # time the run-time plan shape, not the interpreter that covers its compile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quickstep_amd.capi as capi  # noqa: E402
from quickstep_amd import types as T  # noqa: E402
from bench import gen_q1_columns_gpu, q1_config  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
cols = gen_q1_columns_gpu(n, dev, 4)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for label, env in (("plan shape", None), ("interpreter", "1"), ("run-time shape", "jit")):
    # NO_SPECIALIZE must stay set while the state is UPDATED: it also gates the run-time plan shape (an earlier version
    # of this script dropped it after creation and its "interpreter" line timed the hipRTC kernel)
    os.environ.pop("QSX_AGG_NO_SPECIALIZE", None)
    os.environ.pop("QSX_AGG_JIT_MIN_ROWS", None)
    if env == "1":
        os.environ["QSX_AGG_NO_SPECIALIZE"] = "1"
    st = capi.AggState(q1_config()) if env != "jit" else None
    if env == "jit":
        os.environ["QSX_AGG_NO_SPECIALIZE"] = "1"
        st = capi.AggState(q1_config())          # no AOT shape picked at creation ...
        os.environ.pop("QSX_AGG_NO_SPECIALIZE", None)
        os.environ["QSX_AGG_JIT_MIN_ROWS"] = "0"  # ... and the hipRTC shape from the first update
    ms = timed(lambda: st.update(cols, n))
    print(f"Q1 {label:12s} {ms:8.3f} ms  {34 * n / ms / 1e6:8.1f} GB/s  {34 * n / ms / 1e6 / 80:5.1f} % of 8 TB/s")
# Q1 with MIN/MAX instead of two of the sums (no AOT plan shape exists), interpreted
os.environ.pop("QSX_AGG_JIT_MIN_ROWS", None)
os.environ["QSX_AGG_NO_SPECIALIZE"] = "1"
cfg = T.make_agg_config(T.AGG_COMPACT_KEY, [(T.CHAR, 1), (T.CHAR, 1)] + [(T.DOUBLE, None)] * 4, keys=[0, 1],
                        instrs=[(T.EX_SUB, 0, T.const(0), T.col(4)), (T.EX_MUL, 1, T.col(3), T.temp(0))], consts=[1.0],
                        aggs=[(T.AGG_SUM, T.col(2)), (T.AGG_MIN, T.col(3)), (T.AGG_MAX, T.temp(1)), (T.AGG_AVG, T.col(5)),
                              (T.AGG_COUNT_STAR, None)], est_groups=6)
st = capi.AggState(cfg)
ms = timed(lambda: st.update(cols, n))
print(f"Q1-like SUM/MIN/MAX/AVG (interpreter) {ms:8.3f} ms  {34 * n / ms / 1e6:8.1f} GB/s")

# ---- where the interpreter's time goes: variants of the Q1 configuration -----------------------------------------
Q1_COLS = [(T.CHAR, 1), (T.CHAR, 1)] + [(T.DOUBLE, None)] * 4
Q1_INSTRS = [(T.EX_SUB, 0, T.const(0), T.col(4)), (T.EX_MUL, 1, T.col(3), T.temp(0)),
             (T.EX_ADD, 2, T.const(0), T.col(5)), (T.EX_MUL, 3, T.temp(1), T.temp(2))]
variants = {
    "count only": dict(aggs=[(T.AGG_COUNT_STAR, None)]),
    "1 sum(col)": dict(aggs=[(T.AGG_SUM, T.col(2))]),
    "4 sum(col)": dict(aggs=[(T.AGG_SUM, T.col(c)) for c in (2, 3, 4, 5)]),
    "4 sum(col) + 4 instrs unused": dict(aggs=[(T.AGG_SUM, T.col(c)) for c in (2, 3, 4, 5)], instrs=Q1_INSTRS),
    "2 sum(col) + sum(t1) + sum(t3)": dict(aggs=[(T.AGG_SUM, T.col(2)), (T.AGG_SUM, T.col(3)), (T.AGG_SUM, T.temp(1)),
                                                 (T.AGG_SUM, T.temp(3))], instrs=Q1_INSTRS),
}
os.environ["QSX_AGG_NO_SPECIALIZE"] = "1"
for name, kw in variants.items():
    cfg = T.make_agg_config(T.AGG_COMPACT_KEY, Q1_COLS, keys=[0, 1], consts=[1.0], est_groups=6, **kw)
    st = capi.AggState(cfg)
    ms = timed(lambda: st.update(cols, n))
    print(f"interpreter, {name:34s} {ms:8.3f} ms")
