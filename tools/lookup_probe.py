#!/usr/bin/env python3
"""Micro-benchmark of the lookup kernels on Q3-at-SF100-shaped inputs: 600 M clustered l_orderkey values, a 150 M-key
orders side of which ~10 % qualifies, a 54 % input bitmap.  usage: lookup_probe.py [rows_millions]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quickstep_amd.capi as capi  # noqa: E402
from quickstep_amd import types as T  # noqa: E402

dev = torch.device("cuda", 0)
n_o = int(float(sys.argv[1]) * 1e6 / 4) if len(sys.argv) > 1 else 150_000_000
g = torch.Generator(device=dev)
g.manual_seed(1)
lines = torch.randint(1, 8, (n_o,), device=dev, generator=g)
l_orderkey = torch.repeat_interleave(torch.arange(1, n_o + 1, device=dev, dtype=torch.int32), lines)
n = l_orderkey.numel()
o_orderkey = torch.arange(1, n_o + 1, device=dev, dtype=torch.int32)
o_ok, _ = capi.select_cmp(torch.randint(0, 10, (n_o,), device=dev, generator=g, dtype=torch.int32), T.EQ, 3)
l_sel, _ = capi.select_cmp(torch.randint(0, 100, (n,), device=dev, generator=g, dtype=torch.int32), T.LT, 54)
table = capi.JoinTable(T.INT, n_o, key_range=(1, n_o))
table.build(o_orderkey, filter_bitmap=o_ok)
lip = capi.LipFilter(T.LIP_BITVECTOR_EXACT, n_o, 1)
lip.build(o_orderkey, filter_bitmap=o_ok)
l_lip, cnt = lip.probe(l_orderkey, in_bitmap=l_sel)
live = int(cnt.item())
out = (torch.empty(live, dtype=torch.int32, device=dev), torch.empty(live, dtype=torch.int32, device=dev),
       torch.zeros(1, dtype=torch.int64, device=dev))


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


res = {"rows": n, "lip_live": live}
res["select (stream reference)"] = timed(lambda: capi.select_cmp(l_orderkey, T.GT, 5))
res["lip_probe in=54%"] = timed(lambda: lip.probe(l_orderkey, in_bitmap=l_sel))
res["lip_probe no in_bitmap"] = timed(lambda: lip.probe(l_orderkey))
res["probe_count filter=5%"] = timed(lambda: table.probe_count(l_orderkey, filter_bitmap=l_lip))
res["probe filter=5%"] = timed(lambda: table.probe(l_orderkey, capacity=live, filter_bitmap=l_lip, out=out))
res["probe_exists filter=54%"] = timed(lambda: table.probe_exists(l_orderkey, filter_bitmap=l_sel))
res["probe_exists filter=5%"] = timed(lambda: table.probe_exists(l_orderkey, filter_bitmap=l_lip))
res["probe_exists no filter"] = timed(lambda: table.probe_exists(l_orderkey))
res["probe_count filter=54%"] = timed(lambda: table.probe_count(l_orderkey, filter_bitmap=l_sel))
res["probe_count no filter"] = timed(lambda: table.probe_count(l_orderkey))
# C2 shape: 1 M unique build keys, 100 M uniformly random probe keys, every probe row matches
b2 = torch.randperm(1_000_000, device=dev, generator=g, dtype=torch.int32)
p2 = torch.randint(0, 1_000_000, (100_000_000,), device=dev, generator=g, dtype=torch.int32)
t2 = capi.JoinTable(T.INT, 1_000_000, key_range=(0, 999_999))
t2.build(b2)
out2 = (torch.empty(100_000_000, dtype=torch.int32, device=dev), torch.empty(100_000_000, dtype=torch.int32, device=dev),
        torch.zeros(1, dtype=torch.int64, device=dev))
res["C2 probe_count"] = timed(lambda: t2.probe_count(p2))
res["C2 probe"] = timed(lambda: t2.probe(p2, capacity=100_000_000, out=out2))
res["C2 probe_exists"] = timed(lambda: t2.probe_exists(p2))
t3 = capi.JoinTable(T.INT, 1_000_000)          # hashed flavour, same inputs
t3.build(b2)
res["C2 hashed probe_count"] = timed(lambda: t3.probe_count(p2))
res["C2 hashed probe"] = timed(lambda: t3.probe(p2, capacity=100_000_000, out=out2))
res["C2 hashed probe_exists"] = timed(lambda: t3.probe_exists(p2))
print(json.dumps(res))
