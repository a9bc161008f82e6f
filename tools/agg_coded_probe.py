#!/usr/bin/env python3
"""Q1 aggregation over lineitem as the reference's TPC-H DDL stores it (CompressedColumnStore: quantity / discount / tax
dictionary-coded in one byte each) against the same aggregation over plain DOUBLE columns.  usage: agg_coded_probe.py [rows_millions]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("QSX_AGG_JIT_MIN_ROWS", "0")
import quickstep_amd.capi as capi  # noqa: E402
from quickstep_amd import types as T  # noqa: E402
import bench  # noqa: E402

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
qty_d = torch.arange(1, 51, device=dev, dtype=torch.float64)
disc_d = torch.arange(0, 11, device=dev, dtype=torch.float64) / 100
tax_d = torch.arange(0, 9, device=dev, dtype=torch.float64) / 100

plain_cfg = bench.q1_config()
coded_cfg = bench.q1_config()
for c in (2, 4, 5):
    coded_cfg.column_code_width[c] = 1


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


res = {"rows": n}
coded = capi.AggState(coded_cfg)
cols_c = [k1, k2, qty_c, price, disc_c, tax_c]
dicts = [None, None, qty_d, None, disc_d, tax_d]


def run_coded():
    coded.clear()
    coded.update_coded(cols_c, dicts, n)


res["coded_ms (13 B/row)"] = timed(run_coded)
# what the dictionary reads cost: the same launch with every coded column truncation-compressed (value = code, no dictionary)
trunc = capi.AggState(coded_cfg)


def run_truncated():
    trunc.clear()
    trunc.update_coded(cols_c, [None] * 6, n)


res["coded_without_dictionaries_ms"] = timed(run_truncated)
# the same rows as the operators hand them over: a run of 4 MiB blocks (322 640 rows of 13 B), every block with dictionaries
# of its own (qsx_agg_update_coded_blocks_sized; the reference compresses block by block)
block_rows = int(os.environ.get("QSX_PROBE_BLOCK_ROWS", "322640"))
run_blocks, run_dicts = [], []
for lo in range(0, n, block_rows):
    hi = min(n, lo + block_rows)
    run_blocks.append([c[lo:hi] for c in cols_c])
    run_dicts.append([None, None, qty_d.clone(), None, disc_d.clone(), tax_d.clone()])
blocks_state = capi.AggState(coded_cfg)


# (the argument arrays are built once: filling 1860 x 6 ctypes slots per call costs Python more than the launch takes)
import ctypes as C  # noqa: E402
nb, ncols = len(run_blocks), 6
arg_rows = (C.c_int64 * nb)(*[b[0].numel() for b in run_blocks])
arg_cols, arg_dicts, arg_entries = (C.c_void_p * (nb * ncols))(), (C.c_void_p * (nb * ncols))(), (C.c_int32 * (nb * ncols))()
for i in range(nb):
    for c in range(ncols):
        arg_cols[i * ncols + c] = run_blocks[i][c].data_ptr()
        d = run_dicts[i][c]
        arg_dicts[i * ncols + c] = d.data_ptr() if d is not None else None
        arg_entries[i * ncols + c] = d.numel() if d is not None else 0


def update_run(state):
    rc = capi.lib.qsx_agg_update_coded_blocks_sized(state._h, nb, arg_rows, arg_cols, arg_dicts, arg_entries, None, None)
    assert rc == 0, rc


def run_coded_blocks():
    blocks_state.clear()
    update_run(blocks_state)


res["blocks"] = len(run_blocks)
res["coded_blocks_ms"] = timed(run_coded_blocks)
os.environ["QSX_AGG_FACTORED"] = "0"
decoding_blocks = capi.AggState(coded_cfg)


def run_decoding_blocks():
    decoding_blocks.clear()
    update_run(decoding_blocks)


res["coded_blocks_decoding_kernels_ms"] = timed(run_decoding_blocks)
os.environ.pop("QSX_AGG_FACTORED")
bk, bv, _, bg = blocks_state.finalize(dev, capacity=16)
ck, cv, _, cg = coded.finalize(dev, capacity=16)
qty, disc, tax = qty_d[qty_c.long()], disc_d[disc_c.long()], tax_d[tax_c.long()]
del qty_c, disc_c, tax_c
# QSX_PROBE_PLAIN_RUNTIME_SHAPE=1: the plain state takes its run-time shape instead of the AOT one (so that QSX_JIT_OPTIONS
# reach it: tools/agg_coded_exp.sh)
if os.environ.get("QSX_PROBE_PLAIN_RUNTIME_SHAPE") == "1":
    os.environ["QSX_AGG_NO_SPECIALIZE"] = "1"
plain = capi.AggState(plain_cfg)
os.environ.pop("QSX_AGG_NO_SPECIALIZE", None)
cols_p = [k1, k2, qty, price, disc, tax]


def run_plain():
    plain.clear()
    plain.update(cols_p, n)


res["plain_ms (34 B/row)"] = timed(run_plain)
pk, pv, _, pg = plain.finalize(dev, capacity=16)
groups = int(pg.item())
same = int(cg.item()) == groups
for a in range(len(pv)):
    x, y = cv[a][:groups].double(), pv[a][:groups].double()
    order_x, order_y = torch.argsort(ck[0][:groups].long() * 256 + ck[1][:groups].long()), torch.argsort(pk[0][:groups].long() * 256 + pk[1][:groups].long())
    same = same and bool(torch.allclose(x[order_x], y[order_y], rtol=1e-9, atol=0))
res["same_result_as_plain"] = same
same_blocks = int(bg.item()) == groups
for a in range(len(pv)):
    x, y = bv[a][:groups].double(), pv[a][:groups].double()
    order_x, order_y = torch.argsort(bk[0][:groups].long() * 256 + bk[1][:groups].long()), torch.argsort(pk[0][:groups].long() * 256 + pk[1][:groups].long())
    same_blocks = same_blocks and bool(torch.allclose(x[order_x], y[order_y], rtol=1e-9, atol=0))
res["blocks_same_result_as_plain"] = same_blocks
res["coded_GBps_of_codes"] = 13 * n / res["coded_ms (13 B/row)"] / 1e6
res["coded_rows_per_s"] = n / res["coded_ms (13 B/row)"] * 1e3
res["plain_rows_per_s"] = n / res["plain_ms (34 B/row)"] * 1e3
print(json.dumps(res))
