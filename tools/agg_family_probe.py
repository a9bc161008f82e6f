#!/usr/bin/env python3
"""Plans of the AOT family (csrc/agg_family.hpp) that no recording under csrc/jit_shapes/ names, in a process that may not compile
(QSX_AGG_JIT=0): time of the FIRST update of a fresh state and of the following ones, against the interpreter (QSX_AGG_FAMILY=0)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["QSX_AGG_JIT"] = "0"
import quickstep_amd.capi as capi  # noqa: E402
from quickstep_amd import types as T  # noqa: E402


def timed(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(1)
    k_int = torch.randint(0, 120, (n,), device=dev, generator=g, dtype=torch.int32)
    k_chr = torch.randint(65, 70, (n,), device=dev, generator=g, dtype=torch.int32).to(torch.uint8)
    vals = [torch.rand(n, device=dev, generator=g, dtype=torch.float64) for _ in range(3)]
    plans = {
        "int key, 3 sums (20 B/row)": (T.make_agg_config(T.AGG_GENERIC, [(T.DOUBLE, None), (T.DOUBLE, None), (T.INT, None), (T.DOUBLE, None)], keys=[2],
                                                         aggs=[(T.AGG_SUM, T.col(0)), (T.AGG_AVG, T.col(3)), (T.AGG_COUNT_STAR, None), (T.AGG_SUM, T.col(1))], est_groups=128),
                                       [vals[0], vals[1], k_int, vals[2]], 28),
        "char + int keys, 1 sum (13 B/row)": (T.make_agg_config(T.AGG_COMPACT_KEY, [(T.INT, None), (T.DOUBLE, None), (T.CHAR, 1)], keys=[2, 0],
                                                                aggs=[(T.AGG_AVG, T.col(1)), (T.AGG_COUNT_STAR, None)], est_groups=600),
                                              [k_int, vals[0], k_chr], 13),
    }
    for name, (cfg, cols, bytes_per_row) in plans.items():
        out = {"plan": name, "rows": n}
        for family in ("1", "0"):
            os.environ["QSX_AGG_FAMILY"] = family
            st = capi.AggState(cfg)
            first = timed(lambda: st.update(cols, n))
            later = sum(timed(lambda: st.update(cols, n)) for _ in range(3)) / 3
            tag = "family" if family == "1" else "interpreter"
            out[tag] = {"first_update_ms": round(first, 3), "update_ms": round(later, 3), "frac_of_8TBps": round(bytes_per_row * n / later / 1e6 / 8000, 3)}
            st.close()
        print(json.dumps(out))


if __name__ == "__main__":
    main()
