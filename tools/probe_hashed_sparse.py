#!/usr/bin/env python3
"""The hashed INT table over SPARSE keys (no directly addressed shadow possible): C2 shape 1 M x 100 M, match rates 1.0 and
0.2, pairs / count / exists, and clear + build.  One JSON line.  usage: probe_hashed_sparse.py [build_rows]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quickstep_amd.capi as capi  # noqa: E402
from quickstep_amd import types as T  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(2)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
npr = 100_000_000
spread = lambda k: (k.long() * 2039 % (2**31 - 1)).to(torch.int32)   # noqa: E731  (a bijection: unique, sparse keys)
build = spread(torch.randperm(nb, device=dev, generator=g, dtype=torch.int32))
out = (torch.empty(npr, dtype=torch.int32, device=dev), torch.empty(npr, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int64, device=dev))
table = capi.JoinTable(T.INT, nb)
table.build(build)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


res = {"build_rows": nb, "probe_rows": npr}
res["clear_build_ms"] = timed(lambda: (table.clear(), table.build(build)))
for m in (1.0, 0.2):
    probe = spread(torch.randint(0, int(nb / m), (npr,), device=dev, generator=g, dtype=torch.int32))
    res[f"pairs_m{m}_ms"] = timed(lambda: table.probe(probe, capacity=npr, out=out))
    k = int(out[2].item())
    assert bool((build[out[1][:k].long()] == probe[out[0][:k].long()]).all())
    res[f"matches_m{m}"] = k
    res[f"count_m{m}_ms"] = timed(lambda: table.probe_count(probe))
    res[f"exists_m{m}_ms"] = timed(lambda: table.probe_exists(probe))
    del probe
print(json.dumps(res))
