#!/usr/bin/env python3
"""Joins on compressed key stripes (qsx_join_*_blocks_coded): a run of probe blocks whose LONG / INT key lies truncated or
dictionary-coded, probed as it lies, against the same run decoded first (qsx_decode_codes + the plain form) and against plain
value stripes.  usage: python tools/join_coded_probe.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quickstep_amd.capi as capi  # noqa: E402
from quickstep_amd import types as T  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(3)
BLOCKS, ROWS = 100, 1_000_000
N_BUILD = 1_000_000


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for key_type, dtype, label, code_dtype, width, use_dict in (
        (T.LONG, torch.int64, "LONG key truncated to 4 bytes", torch.int32, 4, False),
        (T.INT, torch.int32, "INT key as 2-byte dictionary codes (50 000 values a block)", torch.int16, 2, True),
        (T.LONG, torch.int64, "LONG key as 2-byte dictionary codes (50 000 values a block)", torch.int16, 2, True)):
    table = capi.JoinTable(key_type, N_BUILD, key_range=(0, N_BUILD - 1))
    table.build(torch.randperm(N_BUILD, device=dev, generator=g).to(dtype))
    values, codes, coding = [], [], []
    for b in range(BLOCKS):
        if use_dict:
            d = torch.sort(torch.randperm(N_BUILD + N_BUILD // 4, device=dev, generator=g)[:50_000])[0].to(dtype)
            c = torch.randint(0, 50_000, (ROWS,), device=dev, generator=g, dtype=torch.int32)
            values.append(d[c.long()])
            codes.append(c.to(torch.int16))      # (codes < 2^15 here: the same bytes as the unsigned codes)
            coding.append((2, d))
        else:
            v = torch.randint(0, N_BUILD + N_BUILD // 4, (ROWS,), device=dev, generator=g, dtype=torch.int64)
            values.append(v.to(dtype))
            codes.append(v.to(code_dtype))
            coding.append((width, None))
    want = int(table.probe_count_blocks(values).item())
    assert int(table.probe_count_blocks(codes, coding=coding).item()) == want
    decoded = [torch.empty(ROWS, dtype=dtype, device=dev) for _ in range(BLOCKS)]

    def decode_then_probe():
        for b in range(BLOCKS):
            capi.decode_codes(codes[b], coding[b][1], dtype, out=decoded[b])
        table.probe_count_blocks(decoded)
    n = BLOCKS * ROWS
    res = {"rows": n, "key": label, "matches": want,
           "plain_values_ms": round(timed(lambda: table.probe_count_blocks(values)), 3),
           "coded_as_it_lies_ms": round(timed(lambda: table.probe_count_blocks(codes, coding=coding)), 3),
           "decode_then_probe_ms": round(timed(decode_then_probe, reps=3), 3)}
    ex_coded = timed(lambda: table.probe_exists_blocks(codes, coding=coding))
    ex_plain = timed(lambda: table.probe_exists_blocks(values))   # (after the coded form: the allocator has settled behind the decode buffers)
    res["exists_plain_ms"], res["exists_coded_ms"] = round(ex_plain, 3), round(ex_coded, 3)
    print(json.dumps(res), flush=True)
    del table, values, codes, decoded
