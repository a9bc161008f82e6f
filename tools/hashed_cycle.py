#!/usr/bin/env python3
"""clear + build + first probe of a hashed join table (no statistics: the first probe seals it), the per-query sequence of
bench.py's `hashed_clear_build_probe_ms`, with the phases timed apart by HIP events.  argv: build rows, probe rows, "sparse" for
keys without a dense domain.  Under rocprofv3 --kernel-trace the kernels of the cycle show one by one."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quickstep_amd.capi as capi  # noqa: E402
from quickstep_amd import types as T  # noqa: E402


def main():
    n_build = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
    sparse = len(sys.argv) > 3 and sys.argv[3] == "sparse"
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(2)
    build = torch.randperm(n_build, device=dev, generator=g, dtype=torch.int32)
    probe = torch.randint(0, n_build, (n,), device=dev, generator=g, dtype=torch.int32)
    if sparse:
        spread = lambda k: (k.long() * 2039 % (2**31 - 1)).to(torch.int32)  # noqa: E731
        build, probe = spread(build), spread(probe)
    out = (torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int64, device=dev))
    t = capi.JoinTable(T.INT, n_build)
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
    sums = [0.0, 0.0, 0.0]
    reps = 20
    for it in range(reps + 3):
        e = [ev() for _ in range(4)]
        e[0].record()
        t.clear()
        e[1].record()
        t.build(build)
        e[2].record()
        t.probe(probe, capacity=n, out=out)
        e[3].record()
        torch.cuda.synchronize()
        if it >= 3:
            for i in range(3):
                sums[i] += e[i].elapsed_time(e[i + 1])
    assert int(out[2].item()) == n
    print({"build_rows": n_build, "probe_rows": n, "sparse": sparse, "clear_ms": sums[0] / reps, "build_ms": sums[1] / reps,
           "first_probe_ms": sums[2] / reps, "cycle_ms": sum(sums) / reps})


if __name__ == "__main__":
    main()
