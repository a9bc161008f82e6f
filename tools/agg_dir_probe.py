#!/usr/bin/env python3
"""The group-directory kernels (K8, thousands of groups: one 1024-thread workgroup per CU) over 200 M rows: key box, looked-up
directory, 16-byte key.  For A/B of kernel-source switches through QSX_JIT_OPTIONS (run-time plan shapes, hipRTC build), e.g.
QSX_JIT_COMPILER=hiprtc QSX_JIT_OPTIONS=-DQSX_DMA_ASM=1.  usage: agg_dir_probe.py [rows_millions]"""
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
na = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 200_000_000
g = torch.Generator(device=dev)
g.manual_seed(11)


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


k1 = torch.randint(0, 100, (na,), device=dev, generator=g, dtype=torch.int32)
k2 = torch.randint(0, 100, (na,), device=dev, generator=g, dtype=torch.int32)
val = torch.rand(na, device=dev, generator=g, dtype=torch.float64)
res = {"rows": na, "jit_options": os.environ.get("QSX_JIT_OPTIONS", "")}
gencfg = T.make_agg_config(T.AGG_GENERIC, [(T.INT, None), (T.INT, None), (T.DOUBLE, None)], keys=[0, 1],
                           aggs=[(T.AGG_SUM, T.col(2)), (T.AGG_COUNT_STAR, None)], est_groups=10_000)
st = capi.AggState(gencfg)
res["key_box_ms"] = timed(lambda: st.update([k1, k2, val], na))
ref = st.finalize(dev, capacity=20_000)
groups = int(ref[3].item())
res["groups"] = groups
res["sum_check"] = float(ref[1][0][:groups].double().sum().item())
res["sum_expected"] = float(val.sum().item())
st = capi.AggState(gencfg)
s1, s2 = k1 * 1_000_003, k2 * 7_919 - 11
res["looked_up_ms"] = timed(lambda: st.update([s1, s2, val], na))
del s1, s2
widecfg = T.make_agg_config(T.AGG_GENERIC, [(T.INT, None), (T.LONG, None), (T.INT, None), (T.DOUBLE, None)], keys=[0, 1, 2],
                            aggs=[(T.AGG_SUM, T.col(3)), (T.AGG_COUNT_STAR, None)], est_groups=10_000)
st = capi.AggState(widecfg)
k_long, k_bit = k2.long() << 33, k1 & 1
res["wide_key_ms"] = timed(lambda: st.update([k1, k_long, k_bit, val], na))
print(json.dumps(res))
