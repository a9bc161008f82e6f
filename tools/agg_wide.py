#!/usr/bin/env python3
"""Group-by keys wider than 8 bytes with ~10 k groups: 200 M rows of (INT, LONG, INT) + DOUBLE,
SUM + COUNT(*).  looked_up = the bench_ops shape (components spread: directory entries with the key words); key_box =
component ranges small enough for positional group numbers; three_words = a 24-byte key; narrow = the 2-INT-key reference
points (key box / spread keys).  One JSON line each.  usage: python tools/agg_wide.py [rows]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quickstep_amd.capi as capi  # noqa: E402
from quickstep_amd import types as T  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
os.environ.setdefault("QSX_AGG_JIT_MIN_ROWS", "0")


def timed(fn, reps=3):
    fn()
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


k1 = torch.randint(0, 100, (n,), device=dev, generator=g, dtype=torch.int32)
k2 = torch.randint(0, 100, (n,), device=dev, generator=g, dtype=torch.int32)
val = torch.rand(n, device=dev, generator=g, dtype=torch.float64)
aggs = [(T.AGG_SUM, T.col(3)), (T.AGG_COUNT_STAR, None)]
wide = [(T.INT, None), (T.LONG, None), (T.INT, None), (T.DOUBLE, None)]
shapes = {
    "looked_up": (wide, [k1, k2.long() << 33, k1 & 1, val], 24),
    "key_box": (wide, [k1 >> 1, k2.long() + (1 << 40), k1 & 1, val], 24),
    "three_words": ([(T.LONG, None), (T.LONG, None), (T.LONG, None), (T.DOUBLE, None)],
                    [k1.long() * ((1 << 41) + 3), k2.long() << 33, (k1 & 1).long(), val], 32),
}
for name, (layout, cols, row_bytes) in shapes.items():
    for jit in ("0", str(1 << 60)):
        os.environ["QSX_AGG_JIT_MIN_ROWS"] = jit
        cfg = T.make_agg_config(T.AGG_GENERIC, layout, keys=[0, 1, 2], aggs=aggs, est_groups=10_000)
        st = capi.AggState(cfg)
        ms = timed(lambda: st.update(cols, n))
        groups = int(st.num_groups())
        print(json.dumps({"shape": name, "path": "run-time plan shape" if jit == "0" else "interpreter", "rows": n, "ms": round(ms, 3),
                          "groups": groups, "G_rows_per_s": round(n / ms / 1e6, 1), "GBps": round(row_bytes * n / ms / 1e6, 1)}), flush=True)
        st.close()
    del cols
os.environ["QSX_AGG_JIT_MIN_ROWS"] = "0"
narrow = [(T.INT, None), (T.INT, None), (T.DOUBLE, None)]
for name, cols in (("narrow_key_box", [k1, k2, val]), ("narrow_looked_up", [k1 * 1_000_003, k2 * 7_919 - 11, val])):
    cfg = T.make_agg_config(T.AGG_GENERIC, narrow, keys=[0, 1], aggs=[(T.AGG_SUM, T.col(2)), (T.AGG_COUNT_STAR, None)], est_groups=10_000)
    st = capi.AggState(cfg)
    ms = timed(lambda: st.update(cols, n))
    print(json.dumps({"shape": name, "path": "run-time plan shape", "rows": n, "ms": round(ms, 3), "groups": int(st.num_groups()),
                      "G_rows_per_s": round(n / ms / 1e6, 1), "GBps": round(16 * n / ms / 1e6, 1)}), flush=True)
    st.close()
