#!/usr/bin/env python3
"""TPC-H Q3 on one GPU, composed from the hot-path entry points exactly like tests/test_gpu_q3_pipeline.py, at
SF = argv[1] (default 10: 1.5 M customers, 15 M orders, ~60 M lineitems), with per-phase timings.  Synthetic
TPC-H-shaped columns generated on the device (dense keys, 1-7 lines per order)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quickstep_amd.capi as capi  # noqa: E402
from quickstep_amd import types as T  # noqa: E402

SF = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
# argv[2] = "nolip": no LIP filters — with directly addressed join tables the head array is itself an exact filter, and
# probing it under the predicate bitmap costs less than building + probing a separate bit vector (measured both ways)
USE_LIP = not (len(sys.argv) > 2 and sys.argv[2] == "nolip")
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(7)
n_c, n_o = int(150_000 * SF), int(1_500_000 * SF)
# argv[4] = "types": the reference's own attribute types (benchmarks/tpch/create.sql): c_mktsegment CHAR(10) compared with
# 'BUILDING', o_orderdate / l_shipdate DATE (8-byte DateLit) — instead of 4-byte integer stand-ins
REAL_TYPES = len(sys.argv) > 4 and sys.argv[4] == "types"
DATE, BUILDING = 19950315, 1
c_custkey = torch.randperm(n_c, device=dev, generator=g, dtype=torch.int32) + 1
c_mktsegment = torch.randint(0, 5, (n_c,), device=dev, generator=g, dtype=torch.int32)


def random_dates(n):
    """raw DateLit bytes (year | month << 32 | day << 40), 1992-01-01 .. 1998-12-28"""
    y = torch.randint(1992, 1999, (n,), device=dev, generator=g, dtype=torch.int64)
    m = torch.randint(1, 13, (n,), device=dev, generator=g, dtype=torch.int64)
    d = torch.randint(1, 29, (n,), device=dev, generator=g, dtype=torch.int64)
    return y | (m << 32) | (d << 40)


if REAL_TYPES:
    words = torch.tensor([list(w.ljust(10, b"\0")) for w in (b"AUTOMOBILE", b"BUILDING", b"FURNITURE", b"MACHINERY", b"HOUSEHOLD")],
                         dtype=torch.uint8, device=dev)
    c_mktsegment_char = words[c_mktsegment.long()].contiguous()      # CHAR(10) stripe: n_c x 10 bytes
    DATE_RAW = T.date_raw(1995, 3, 15)
o_orderkey = torch.randperm(n_o, device=dev, generator=g, dtype=torch.int32) + 1
o_custkey = torch.randint(1, n_c + 1, (n_o,), device=dev, generator=g, dtype=torch.int32)
o_orderdate = random_dates(n_o) if REAL_TYPES else torch.randint(19920101, 19981231, (n_o,), device=dev, generator=g, dtype=torch.int32)
o_shippriority = torch.zeros(n_o, device=dev, dtype=torch.int32)
lines = torch.randint(1, 8, (n_o,), device=dev, generator=g)
l_orderkey = torch.repeat_interleave(torch.arange(1, n_o + 1, device=dev, dtype=torch.int32), lines)   # clustered on orderkey
n_l = l_orderkey.numel()
l_extendedprice = torch.rand(n_l, device=dev, generator=g, dtype=torch.float64) * 104100 + 900
l_discount = torch.randint(0, 11, (n_l,), device=dev, generator=g).double() / 100
l_shipdate = random_dates(n_l) if REAL_TYPES else torch.randint(19920101, 19981231, (n_l,), device=dev, generator=g, dtype=torch.int32)
torch.cuda.synchronize()

cfg = T.make_agg_config(T.AGG_COLLISION_FREE, [(T.INT, None), (T.DOUBLE, None), (T.DOUBLE, None)], keys=[0],
                        instrs=[(T.EX_SUB, 0, T.const(0), T.col(2)), (T.EX_MUL, 1, T.col(1), T.temp(0))],
                        consts=[1.0], aggs=[(T.AGG_SUM, T.temp(1))], num_entries=n_o + 1)
state = capi.AggState(cfg)
# argv[3] = "fused": no materialised join output — the aggregation reads l_orderkey / l_extendedprice / l_discount THROUGH the
# pair list: a "compressed attribute" whose 4-byte codes are the probe tids and whose dictionary is the column itself
FUSED = len(sys.argv) > 3 and sys.argv[3] == "fused"
if FUSED:
    cfg_fused = T.make_agg_config(T.AGG_COLLISION_FREE, [(T.INT, None), (T.DOUBLE, None), (T.DOUBLE, None)], keys=[0],
                                  instrs=[(T.EX_SUB, 0, T.const(0), T.col(2)), (T.EX_MUL, 1, T.col(1), T.temp(0))],
                                  consts=[1.0], aggs=[(T.AGG_SUM, T.temp(1))], num_entries=n_o + 1, code_widths=[4, 4, 4])
    state = capi.AggState(cfg_fused)
t_c = capi.JoinTable(T.INT, n_c, key_range=(1, n_c))
t_o = capi.JoinTable(T.INT, n_o, key_range=(1, n_o))
lip_c = capi.LipFilter(T.LIP_BITVECTOR_EXACT, n_c, 1)
lip_o = capi.LipFilter(T.LIP_BITVECTOR_EXACT, n_o, 1)
os.environ.setdefault("QSX_AGG_JIT_MIN_ROWS", "0")
phases = {}


def run(timed):
    ev = []

    def mark(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        ev.append((name, e))

    mark("start")
    t_c.clear(); t_o.clear(); state.clear()
    # (the LIP filters are rebuilt too: new filter objects are cheap, clear = recreate is avoided by OR-ing the same bits)
    c_sel, _ = capi.select_cmp_char(c_mktsegment_char, T.EQ, b"BUILDING") if REAL_TYPES else capi.select_cmp(c_mktsegment, T.EQ, BUILDING)
    t_c.build(c_custkey, filter_bitmap=c_sel)
    if USE_LIP:
        lip_c.build(c_custkey, filter_bitmap=c_sel)
    mark("customer: select + build + LIP build")
    o_sel, _ = capi.select_cmp(o_orderdate, T.LT, DATE_RAW, qtype=T.DATE) if REAL_TYPES else capi.select_cmp(o_orderdate, T.LT, DATE)
    o_lip = lip_c.probe(o_custkey, in_bitmap=o_sel)[0] if USE_LIP else o_sel
    o_ok, o_cnt = t_c.probe_exists(o_custkey, filter_bitmap=o_lip)
    t_o.build(o_orderkey, filter_bitmap=o_ok)
    if USE_LIP:
        lip_o.build(o_orderkey, filter_bitmap=o_ok)
    mark("orders: select + LIP probe + semi probe + build + LIP build")
    l_sel, l_sel_count = capi.select_cmp(l_shipdate, T.GT, DATE_RAW, qtype=T.DATE) if REAL_TYPES else capi.select_cmp(l_shipdate, T.GT, DATE)
    mark("lineitem: select l_shipdate > DATE")
    if USE_LIP:
        l_lip, l_live = lip_o.probe(l_orderkey, in_bitmap=l_sel)
    else:
        l_lip, l_live = l_sel, l_sel_count
    mark("lineitem: LIP probe on l_orderkey")
    # o_orderkey is the primary key of the build side: at most one match per probe row, so the rows that pass the LIP
    # filter bound the output (the reference sizes from the same uniqueness fact, impliesUniqueAttributes)
    p, b, cnt = t_o.probe(l_orderkey, capacity=int(l_live.item()), filter_bitmap=l_lip)
    total = int(cnt.item())
    mark("lineitem: inner probe")
    if FUSED:
        mark("gather 3 lineitem columns")
        pt = p[:total]
        state.update_coded([pt, pt, pt], [l_orderkey, l_extendedprice, l_discount], total)
    else:
        key = capi.gather(l_orderkey, p[:total])
        price = capi.gather(l_extendedprice, p[:total])
        disc = capi.gather(l_discount, p[:total])
        mark("gather 3 lineitem columns")
        state.update([key, price, disc], total)
    keys, vals, nulls, groups = state.finalize(dev)
    gcount = int(groups.item())
    mark("dense group-by SUM(price*(1-disc)) + finalize")
    perm = capi.sort_top_k([vals[0][:gcount]], 10, [True])
    top_keys = capi.gather(keys[0][:gcount], perm)
    top_rev = capi.gather(vals[0][:gcount], perm)
    mark("ORDER BY revenue DESC LIMIT 10")
    torch.cuda.synchronize()
    if timed:
        for (_, a), (name, e) in zip(ev[:-1], ev[1:]):
            phases[name] = phases.get(name, 0.0) + a.elapsed_time(e)
    return total, gcount, top_keys, top_rev


run(False)
reps = 3
t0 = time.perf_counter()
for _ in range(reps):
    pairs, groups, top_keys, top_rev = run(True)
wall = (time.perf_counter() - t0) / reps * 1e3
rows = n_c + n_o + n_l
print(json.dumps({"query": "TPC-H Q3 (synthetic, 1 GPU)" + ("" if USE_LIP else ", no LIP filters") + (", aggregation through the pair list" if FUSED else "") + (", CHAR(10) / DATE attributes" if REAL_TYPES else ""), "SF": SF, "customer": n_c, "orders": n_o, "lineitem": n_l, "joined_pairs": pairs,
                  "groups": groups, "wall_ms": wall, "input_rows_per_s": rows / wall * 1e3,
                  "phases_ms": {k: v / reps for k, v in phases.items()}, "top_revenue": top_rev.cpu().tolist()[:3]}))
