#!/usr/bin/env python3
"""K9 over a run of blocks (qsx_partition_scatter_blocks) at C4's shape: lineitem's 75 M rows as 215 blocks of 349 525 rows
(4 MB of the widest attribute), INT key + 8-byte payload into P partitions.  Three ways to the same scattered columns, checked
equal: the scatter over one stripe (the rows already end to end), the scatter that reads the blocks where they lie, and the
blocks copied end to end first (qsx_copy_segments) and scattered then — what the repartitioning Select did.
argv: rows, rows per block, P."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quickstep_amd.capi as capi  # noqa: E402


def timed(fn, reps=10):
    fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 75_019_492
    per_block = int(sys.argv[2]) if len(sys.argv) > 2 else 349_525
    P = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    cuts = list(range(0, n, per_block)) + [n]
    # every block an allocation of its own, as the storage manager's blocks are
    keys = [torch.randint(0, 2**31 - 1, (b - a,), device=dev, generator=g, dtype=torch.int32) for a, b in zip(cuts[:-1], cuts[1:])]
    pays = [k.long() * 3 + 1 for k in keys]
    all_keys, all_pays = torch.cat(keys), torch.cat(pays)
    out = {"rows": n, "blocks": len(keys), "rows_per_block": per_block, "P": P}
    want = [None]
    got = [None]
    staged = [None]

    def one_stripe():
        want[0] = capi.partition_scatter(all_keys, P, [all_keys, all_pays])

    def blocks():
        got[0] = capi.partition_scatter_blocks(keys, P, [[k, p] for k, p in zip(keys, pays)])

    stage_k, stage_p = torch.empty_like(all_keys), torch.empty_like(all_pays)

    def copy_then_scatter():
        capi.copy_segments(keys + pays, [stage_k[a:b] for a, b in zip(cuts[:-1], cuts[1:])] + [stage_p[a:b] for a, b in zip(cuts[:-1], cuts[1:])])
        staged[0] = capi.partition_scatter(stage_k, P, [stage_k, stage_p])

    for name, fn in (("one_stripe_ms", one_stripe), ("blocks_where_they_lie_ms", blocks), ("copy_then_scatter_ms", copy_then_scatter)):
        out[name] = round(timed(fn), 4)
    for other in (got[0], staged[0]):
        assert torch.equal(other[1], want[0][1])
        for a, b in zip(other[0], want[0][0]):
            assert torch.equal(a, b)
    out["checked"] = True
    moved = 12 * 2 * n
    out["blocks_GBps_moved"] = round(moved / out["blocks_where_they_lie_ms"] / 1e6, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
