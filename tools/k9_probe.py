#!/usr/bin/env python3
"""K9 (qsx_partition_scatter) at the shuffle's shapes: 100 M rows into P partitions, (key, tid) and C4's (key, 8-byte payload).
Checks the result (offsets = counts, every row in its partition, tuple ids ascending inside a partition = stable) and
prints ms per call by HIP events.  argv: rows, P.  QSX_K9_WAVES=0: the workgroup-tile kernels of rounds 3-5."""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quickstep_amd.capi as capi  # noqa: E402


def timed(fn, reps=5):
    fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    keys = torch.randint(0, 2**31 - 1, (n,), device=dev, generator=g, dtype=torch.int32)
    tids = torch.arange(n, device=dev, dtype=torch.int32)
    pay = keys.long() * 3 + 1
    out = {"rows": n, "P": P, "waves": os.environ.get("QSX_K9_WAVES", "1")}
    for name, cols, bytes_moved in (("key_tid", [keys, tids], 8 * 2), ("key_payload8", [keys, pay], 12 * 2), ("key_only", [keys], 4 * 2)):
        lib = capi.lib
        widths = (C.c_int32 * len(cols))(*[c.element_size() for c in cols])
        outs = [torch.empty_like(c) for c in cols]
        ws_bytes = lib.qsx_partition_workspace_bytes(n, P)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        offsets = torch.zeros(P + 1, dtype=torch.int64, device=dev)
        src = (C.c_void_p * len(cols))(*[c.data_ptr() for c in cols])
        dst = (C.c_void_p * len(cols))(*[c.data_ptr() for c in outs])

        def call():
            rc = lib.qsx_partition_scatter(capi.qsx_type_of(keys), C.c_void_p(keys.data_ptr()), n, P, len(cols), src, widths, dst,
                                           C.c_void_p(offsets.data_ptr()), C.c_void_p(ws.data_ptr()), ws_bytes, None)
            assert rc == 0, rc
        ms = timed(call)
        if os.environ.get("QSX_EXP_K9"):
            out[name] = {"ms": round(ms, 4)}
            continue
        off = offsets.tolist()
        pid = (keys.long() & 0xFFFFFFFF) % P
        counts = torch.bincount(pid, minlength=P).tolist()
        assert [off[i + 1] - off[i] for i in range(P)] == counts and off[0] == 0 and off[P] == n, (off, counts)
        ok = outs[0].long() & 0xFFFFFFFF
        for p in range(P):
            seg = slice(off[p], off[p + 1])
            assert bool(((ok[seg] % P) == p).all()), f"partition {p}: a row of another partition"
            if name == "key_tid":
                t = outs[1][seg]
                assert bool((t[1:] > t[:-1]).all()), f"partition {p}: not stable"
                assert bool((keys[t.long()] == outs[0][seg]).all())
            if name == "key_payload8":
                assert bool((outs[1][seg] == outs[0][seg].long() * 3 + 1).all())
        out[name] = {"ms": round(ms, 4), "GBps_moved": round(bytes_moved * n / ms / 1e6, 1), "frac_of_8TBps": round(bytes_moved * n / ms / 1e6 / 8000, 3)}
        del outs, ws
    print(json.dumps(out))


if __name__ == "__main__":
    main()
