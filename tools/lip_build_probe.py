#!/usr/bin/env python3
"""K12 build (qsx_lip_build) of an exact bit vector over TPC-H Q3's orders range (56.25 M keys): 5.56 M qualifying keys in key
order (what dbgen writes), the same keys shuffled (bench.py's synthetic relations), and every key of the range in order.
ms per build by HIP events, the clear of the filter subtracted."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quickstep_amd.capi as capi  # noqa: E402
from quickstep_amd import types as T  # noqa: E402


def timed(fn, reps=20):
    fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    card = 56_250_000
    some = torch.randperm(card, device=dev, generator=g, dtype=torch.int32)[:5_560_000] + 1
    out = {"cardinality": card}
    for name, keys in (("keys_in_order_one_in_ten_ms", torch.sort(some)[0]), ("keys_shuffled_one_in_ten_ms", some),
                       ("every_key_in_order_ms", torch.arange(1, card + 1, device=dev, dtype=torch.int32))):
        f = capi.LipFilter(T.LIP_BITVECTOR_EXACT, card, 1)

        def run():
            f.clear()
            f.build(keys)
        out[name] = round(timed(run) - timed(f.clear), 4)
        out[name.replace("_ms", "_keys")] = int(keys.numel())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
