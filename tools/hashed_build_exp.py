#!/usr/bin/env python3
"""Times qsx_join_build on a hashed table (1 M INT keys) alone; QSX_EXP_BUILD / QSX_EXP_BUILD_GRID pick experiment kernels."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quickstep_amd.capi as capi  # noqa: E402
from quickstep_amd import types as T  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(2)
build = torch.randperm(n, device=dev, generator=g, dtype=torch.int32)
t = capi.JoinTable(T.INT, n)
tot = 0.0
for it in range(23):
    t.clear()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); t.build(build); b.record(); torch.cuda.synchronize()
    if it >= 3: tot += a.elapsed_time(b)
print(os.environ.get("QSX_EXP_BUILD", "0"), os.environ.get("QSX_EXP_BUILD_GRID", "-"), "build_ms", round(tot / 20, 4))
