#!/usr/bin/env python3
"""K4 against SMALL build sides: 100 M probe keys (match rate 1.0) of a table of 25 K / 8 K / 2 K / 30 key values — the
probe from LDS (csrc/join_lds.hpp) against the same table looked up in L2 (QSX_JOIN_LDS=0).  dense = exact statistics,
hashed = no statistics (dense keys: the shadow; sparse keys: the bucketed table).  usage: probe_small_tables.py [probe_millions]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quickstep_amd.capi as capi  # noqa: E402
from quickstep_amd import types as T  # noqa: E402

dev = torch.device("cuda", 0)
n = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 100_000_000
g = torch.Generator(device=dev)
g.manual_seed(3)
out = (torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int64, device=dev))


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


spread = lambda k: (k.long() * 2039 % (2**31 - 1)).to(torch.int32)   # noqa: E731
for keys_n in (25_000, 8_000, 2_000, 30):
    build = torch.randperm(keys_n, device=dev, generator=g, dtype=torch.int32)
    probe = torch.randint(0, keys_n, (n,), device=dev, generator=g, dtype=torch.int32)
    for kind in ("dense", "hashed", "hashed_sparse"):
        bk, pk = (spread(build), spread(probe)) if kind == "hashed_sparse" else (build, probe)
        res = {"build_keys": keys_n, "probe_rows": n, "table": kind}
        for lds in ("1", "0"):
            os.environ["QSX_JOIN_LDS"] = lds
            t = capi.JoinTable(T.INT, keys_n, key_range=(0, keys_n - 1) if kind == "dense" else None)
            t.build(bk)
            ms = timed(lambda: t.probe(pk, capacity=n, out=out))
            k = int(out[2].item())
            assert k == n and bool((bk[out[1][:k].long()] == pk[out[0][:k].long()]).all())
            tag = "lds" if lds == "1" else "l2"
            res[f"pairs_ms_{tag}"] = ms
            res[f"count_ms_{tag}"] = timed(lambda: t.probe_count(pk))
            res[f"frac_of_hbm_peak_{tag}"] = (4 * n + 8 * k) / ms / 1e6 / 8000.0
            t.close()
        print(json.dumps(res), flush=True)
os.environ.pop("QSX_JOIN_LDS", None)
