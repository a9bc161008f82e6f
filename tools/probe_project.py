#!/usr/bin/env python3
"""Probe + projection (qsx_join_probe_project_blocks) against probe + gathers, 100 M probe rows of a 1 M-key directly addressed
table: one INT attribute from each side (the operators' bench's output relation), one LONG from each side, with the covering
array and without (QSX_JOIN_COVER=0).  One JSON line.  usage: python tools/probe_project.py [probe_rows] [build_rows]"""
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
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return round(a.elapsed_time(b) / reps, 3)


bk = torch.randperm(nb, device=dev, generator=g, dtype=torch.int32)
pk = torch.randint(0, nb, (n,), device=dev, generator=g, dtype=torch.int32)
pay_b, pay_p = bk.long() * 3 + 1, pk.long() * 7 + 2
b_attr = bk * 2 + 1
line = {"probe_rows": n, "build_rows": nb}
for cover in ("1", "0"):
    os.environ["QSX_JOIN_COVER"] = cover
    t = capi.JoinTable(T.INT, nb, key_range=(0, nb - 1))
    t.build(bk)
    tag = "cover" if cover == "1" else "head_and_stripes"
    line[f"project_int_int_ms_{tag}"] = timed(lambda: t.probe_project_blocks([pk], [[pk]], [[b_attr]], capacity=n))
    line[f"project_long_long_ms_{tag}"] = timed(lambda: t.probe_project_blocks([pk], [[pay_p]], [[pay_b]], capacity=n))
    line[f"project_int_only_probe_side_ms_{tag}"] = timed(lambda: t.probe_project_blocks([pk], [[pk]], [], capacity=n))
    if cover == "1":
        out = (torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int64, device=dev))
        line["pairs_ms"] = timed(lambda: t.probe(pk, capacity=n, out=out))
        ob, op = torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev)

        def unfused():
            t.probe(pk, capacity=n, out=out)
            capi.gather(b_attr, out[1], out=ob)
            capi.gather(pk, out[0], out=op)
        line["pairs_and_two_int_gathers_ms"] = timed(unfused)
    t.close()
print(json.dumps(line))
