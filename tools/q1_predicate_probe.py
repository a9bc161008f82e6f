#!/usr/bin/env python3
"""TPC-H Q1 as the reference runs it — the l_shipdate predicate INSIDE the aggregation — over lineitem's code stripes
(13 B/row + 4 B/row of l_shipdate): the factored kernels behind the predicate pass against the decoding kernels
(QSX_AGG_FACTORED=0), one stripe and a run of 4 MiB blocks.  usage: q1_predicate_probe.py [rows_millions]"""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("QSX_AGG_JIT_MIN_ROWS", "0")
import quickstep_amd.capi as capi  # noqa: E402
from quickstep_amd import types as T  # noqa: E402

dev = torch.device("cuda", 0)
n = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 600_000_000
g = torch.Generator(device=dev)
g.manual_seed(4)
combo = torch.multinomial(torch.tensor([0.2466, 0.0065, 0.5005, 0.2464], device=dev), n, replacement=True, generator=g)
k1 = torch.tensor(list(b"ANNR"), dtype=torch.uint8, device=dev)[combo]
k2 = torch.tensor(list(b"FFOF"), dtype=torch.uint8, device=dev)[combo]
del combo
qty_c = torch.randint(0, 50, (n,), device=dev, generator=g, dtype=torch.uint8)
disc_c = torch.randint(0, 11, (n,), device=dev, generator=g, dtype=torch.uint8)
tax_c = torch.randint(0, 9, (n,), device=dev, generator=g, dtype=torch.uint8)
price = (torch.rand(n, device=dev, generator=g, dtype=torch.float64) * 104100 + 900).mul(100).round().div(100)
ship = torch.randint(19920101, 19920101 + 2526, (n,), device=dev, generator=g, dtype=torch.int32)   # ~98 % pass, as in Q1
cutoff = 19920101 + 2475
qty_d = torch.arange(1, 51, device=dev, dtype=torch.float64)
disc_d = torch.arange(0, 11, device=dev, dtype=torch.float64) / 100
tax_d = torch.arange(0, 9, device=dev, dtype=torch.float64) / 100


def config():
    cfg = T.make_agg_config(
        T.AGG_COMPACT_KEY, columns=[(T.CHAR, 1), (T.CHAR, 1)] + [(T.DOUBLE, None)] * 4 + [(T.INT, None)], keys=[0, 1],
        instrs=[(T.EX_SUB, 0, T.const(0), T.col(4)), (T.EX_MUL, 1, T.col(3), T.temp(0)), (T.EX_ADD, 2, T.const(0), T.col(5)),
                (T.EX_MUL, 3, T.temp(1), T.temp(2))], consts=[1.0],
        aggs=[(T.AGG_SUM, T.col(2)), (T.AGG_SUM, T.col(3)), (T.AGG_SUM, T.temp(1)), (T.AGG_SUM, T.temp(3)), (T.AGG_AVG, T.col(2)),
              (T.AGG_AVG, T.col(3)), (T.AGG_AVG, T.col(4)), (T.AGG_COUNT_STAR, None)], est_groups=6, pred=[(6, T.LE, cutoff)])
    for c in (2, 4, 5):
        cfg.column_code_width[c] = 1
    return cfg


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


cols = [k1, k2, qty_c, price, disc_c, tax_c, ship]
dicts = [None, None, qty_d, None, disc_d, tax_d, None]
block_rows = 246_672                       # a 4 MiB block of 17-byte tuples (multiple of 16 rows)
starts = list(range(0, n, block_rows))
nb, ncols = len(starts), 7
own = [[None if d is None else d.clone() for d in dicts] for _ in starts]
a_rows = (C.c_int64 * nb)(*[min(n, lo + block_rows) - lo for lo in starts])
a_cols, a_dicts, a_entries = (C.c_void_p * (nb * ncols))(), (C.c_void_p * (nb * ncols))(), (C.c_int32 * (nb * ncols))()
for b, lo in enumerate(starts):
    for c in range(ncols):
        a_cols[b * ncols + c] = cols[c].data_ptr() + lo * cols[c].element_size()
        d = own[b][c]
        a_dicts[b * ncols + c] = d.data_ptr() if d is not None else None
        a_entries[b * ncols + c] = d.numel() if d is not None else 0
res = {"rows": n, "blocks": nb, "selectivity": float((ship <= cutoff).float().mean().item())}
results = {}
for factored in ("1", "0"):
    os.environ["QSX_AGG_FACTORED"] = factored
    st, sb = capi.AggState(config()), capi.AggState(config())

    def one():
        st.clear()
        st.update_coded(cols, dicts, n)

    def run():
        sb.clear()
        rc = capi.lib.qsx_agg_update_coded_blocks_sized(sb._h, nb, a_rows, a_cols, a_dicts, a_entries, None, None)
        assert rc == 0, rc
    name = "factored behind the predicate pass" if factored == "1" else "decoding kernels"
    res[f"one stripe, {name}, ms"] = timed(one)
    res[f"run of blocks, {name}, ms"] = timed(run)
    k, v, _, cnt = st.finalize(dev, capacity=16)
    gcount = int(cnt.item())
    order = torch.argsort(k[0][:gcount].long() * 256 + k[1][:gcount].long())
    kb, vb, _, cntb = sb.finalize(dev, capacity=16)
    orderb = torch.argsort(kb[0][:gcount].long() * 256 + kb[1][:gcount].long())
    results[factored] = [x[:gcount][order].double() for x in v]
    res[f"run equals stripe ({name})"] = all(bool(torch.allclose(x[:gcount][orderb].double(), y, rtol=1e-9, atol=0)) for x, y in zip(vb, results[factored]))
os.environ.pop("QSX_AGG_FACTORED")
res["factored equals decoding"] = all(bool(torch.allclose(x, y, rtol=1e-9, atol=0)) for x, y in zip(results["1"], results["0"]))
res["count equal"] = bool(torch.equal(results["1"][7], results["0"][7]))
print(json.dumps(res))
