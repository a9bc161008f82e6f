#!/usr/bin/env python3
"""Times the join probe variants on the C2 shape (1 M build x 100 M probe):
pairs / count-only / exists through the hashed table (QSX_JOIN_ADAPTIVE=0 for the hashed kernels themselves)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quickstep_amd.capi as capi  # noqa: E402
from quickstep_amd import types as T  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(2)
nb, npr = 1_000_000, 100_000_000
build = torch.randperm(nb, device=dev, generator=g, dtype=torch.int32)
match = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
probe = torch.randint(0, int(nb / match), (npr,), device=dev, generator=g, dtype=torch.int32)
table = capi.JoinTable(T.INT, nb)
table.build(build)
dense = capi.JoinTable(T.INT, nb, key_range=(0, nb - 1))
dense.build(build)
out = (torch.empty(npr, dtype=torch.int32, device=dev), torch.empty(npr, dtype=torch.int32, device=dev),
       torch.zeros(1, dtype=torch.int64, device=dev))


def timed(name, fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:32s} {e0.elapsed_time(e1) / reps:8.3f} ms")


timed("dense build (clear + 1 M)", lambda: (dense.clear(), dense.build(build)))
timed("hashed build (clear + 1 M)", lambda: (table.clear(), table.build(build)))
timed("dense probe pairs", lambda: dense.probe(probe, capacity=npr, out=out))
timed("dense probe count only", lambda: dense.probe_count(probe))
timed("dense probe exists bitmap", lambda: dense.probe_exists(probe))
timed("probe pairs", lambda: table.probe(probe, capacity=npr, out=out))
timed("probe count only", lambda: table.probe_count(probe))
timed("probe exists bitmap", lambda: table.probe_exists(probe))
src = torch.empty(npr, dtype=torch.int32, device=dev)
timed("torch copy 400 MB + 800 MB out", lambda: (src.copy_(probe), out[0].copy_(probe), out[1].copy_(probe)))
