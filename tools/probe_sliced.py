#!/usr/bin/env python3
"""Probe of join tables by table kind and size, 100 M probe keys, pairs and count-only, match rate 1.0 and 0.2:
  dense                the directly addressed table (exact statistics from the optimizer)
  hashed               the hashed table over the same dense keys: the first probe gives it a directly addressed shadow
  hashed_no_shadow     the same with QSX_JOIN_ADAPTIVE=0: the hashed kernels themselves
  hashed_sparse_keys   keys spread over the INT range (no shadow possible)
each with the plain kernels and with the XCD-sliced, compacting kernel forced (QSX_JOIN_SLICED=1, join_sliced.hpp).
One JSON line per (table kind, build rows).  usage: python tools/probe_sliced.py [probe_rows] [build_rows ...]"""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
sizes = [int(a) for a in sys.argv[2:]] or [1_000_000, 2_000_000, 4_000_000, 8_000_000, 16_000_000]


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


out = (torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev),
       torch.zeros(1, dtype=torch.int64, device=dev))
for nb in sizes:
    b = torch.randperm(nb, device=dev, generator=g, dtype=torch.int32)
    # sparse keys for the hashed table: the same permutation spread over the INT range by an odd multiplier (unique, no
    # statistics that would allow direct addressing)
    spread = (b.long() * 2039 % (2**31 - 1)).to(torch.int32)
    p10 = torch.randint(0, nb, (n,), device=dev, generator=g, dtype=torch.int32)
    p02 = torch.randint(0, 5 * nb, (n,), device=dev, generator=g, dtype=torch.int32)
    for kind in ("dense", "hashed", "hashed_no_shadow", "hashed_sparse_keys"):
        build = spread if kind == "hashed_sparse_keys" else b
        os.environ["QSX_JOIN_ADAPTIVE"] = "0" if kind == "hashed_no_shadow" else "1"
        t = capi.JoinTable(T.INT, nb, key_range=(0, nb - 1) if kind == "dense" else None)
        t.build(build)
        line = {"table": kind, "build_keys": nb, "table_MiB": nb * (4 if kind == "dense" else 16) / 2**20, "probe_rows": n}
        for m, probe in (("m1.0", p10), ("m0.2", p02)):
            if kind == "hashed_sparse_keys":
                probe = (probe.long() * 2039 % (2**31 - 1)).to(torch.int32)
            for sliced in ("0", "1"):
                if kind == "hashed" and sliced == "1":
                    continue            # (the shadow is a dense table: see the dense line)
                os.environ["QSX_JOIN_SLICED"] = sliced
                tag = "plain" if sliced == "0" else "sliced"
                line[f"pairs_ms_{m}_{tag}"] = round(timed(lambda: t.probe(probe, capacity=n, out=out)), 3)
                line[f"matches_{m}"] = int(out[2].item())
                line[f"count_ms_{m}_{tag}"] = round(timed(lambda: t.probe_count(probe)), 3)
        print(json.dumps(line), flush=True)
        t.close()
