#!/usr/bin/env python3
"""Calls K9 (qsx_partition_scatter) a few times on 100 M (key, tid) rows; run under rocprofv3 --kernel-trace to
see the per-kernel split (hist / scan / scatter)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quickstep_amd.capi as capi  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
g = torch.Generator(device=dev)
g.manual_seed(1)
keys = torch.randint(0, 1 << 30, (n,), device=dev, generator=g, dtype=torch.int32)
tids = torch.arange(n, device=dev, dtype=torch.int32)
for _ in range(4):
    capi.partition_scatter(keys, P, [keys, tids])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    capi.partition_scatter(keys, P, [keys, tids])
e1.record()
torch.cuda.synchronize()
print(f"partition_scatter P={P}: {e0.elapsed_time(e1) / 5:.3f} ms per call (includes output allocation)")
