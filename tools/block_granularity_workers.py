#!/usr/bin/env python3
"""Work orders per 120 K-row block issued the way the reference issues them: W worker threads, each with its own stream,
pulling blocks from a shared queue (query_execution/Worker.cpp:54-99) — the Q1 aggregation over 120 M rows as 1000 calls.
usage: python tools/block_granularity_workers.py [rows] [block_rows]"""
import json
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quickstep_amd.capi as capi  # noqa: E402
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120_000_000
block = int(sys.argv[2]) if len(sys.argv) > 2 else 120_000
dev = torch.device("cuda:0")
cols = bench.gen_q1_columns_gpu(n, dev, 4)
st = capi.AggState(bench.q1_config())
starts = list(range(0, n, block))
slices = [[c[s:min(n, s + block)] for c in cols] for s in starts]     # views prepared up front: the binding's slicing is not the subject


def run(workers):
    st.clear()
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev) for _ in range(workers)]
    next_block = [0]
    lock = threading.Lock()

    def worker(w):
        s = streams[w]
        while True:
            with lock:
                i = next_block[0]
                next_block[0] += 1
            if i >= len(starts):
                return
            st.update(slices[i], slices[i][0].numel(), stream=s)

    t0 = time.perf_counter()
    threads = [threading.Thread(target=worker, args=(w,)) for w in range(workers)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    issued = time.perf_counter() - t0
    torch.cuda.synchronize()
    return issued * 1e3, (time.perf_counter() - t0) * 1e3


for workers in (1, 2, 4, 8, 16):
    run(workers)
    best = min((run(workers) for _ in range(3)), key=lambda r: r[1])
    keys, vals, _, groups = st.finalize(dev, capacity=16)
    print(json.dumps({"rows": n, "block_rows": block, "calls": len(starts), "workers": workers, "issue_ms": round(best[0], 2),
                      "wall_ms": round(best[1], 2), "count_check": int(vals[7][:int(groups.item())].sum().item()) == n}), flush=True)
