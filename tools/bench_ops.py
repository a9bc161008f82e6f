#!/usr/bin/env python3
"""Per-operator micro-benchmarks of every C-ABI kernel family with achieved
GB/s against algorithmic bytes.  Complements bench.py (which times the headline C2 + C3 step).
usage: python tools/bench_ops.py [scale]   (scale 1.0 = 100 M-row inputs)"""
import json
import os
import sys

import torch

os.environ.setdefault("QSX_AGG_JIT_SYNC", "1")   # time the run-time plan shape, not the interpreter that covers its compile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quickstep_amd.capi as capi  # noqa: E402
from quickstep_amd import types as T  # noqa: E402
from bench import gen_q1_columns_gpu, q1_config  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
N = int(100_000_000 * scale)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(1)
PEAK = 8000.0


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


def report(name, ms, rows, algo_bytes, note=""):
    gbs = algo_bytes / ms / 1e6
    print(json.dumps({"op": name, "ms": round(ms, 4), "rows": rows, "G_rows_per_s": round(rows / ms / 1e6, 2),
                      "algorithmic_GB": round(algo_bytes / 1e9, 3), "achieved_GBps": round(gbs, 1),
                      "frac_of_8TBps": round(gbs / PEAK, 4), "note": note}))


# ---- K1 / K2 select (C1 shape on the GPU) -----------------------------------------------------
col = torch.randint(0, 2**31 - 1, (N,), device=dev, generator=g, dtype=torch.int32)
bm = capi.new_bitmap(N, dev)
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
for k, sel in ((21474836, 0.01), (214748364, 0.10), (1073741824, 0.50)):
    ms = timed(lambda: capi.select_cmp(col, T.LT, k, out_bitmap=bm, out_count=cnt))
    report(f"K1 select_cmp INT col<K sel={sel}", ms, N, 4 * N)
    out = [torch.empty_like(col)]
    ms = timed(lambda: capi.compact_gather([col], bm, N, out_cols=out))
    report(f"K2 compact_gather 1 INT col sel={sel}", ms, N, int(4 * N * sel) * 2 + N // 8, "reads touched lines + writes selected")
dcol = torch.rand(N, device=dev, generator=g, dtype=torch.float64)
ms = timed(lambda: capi.select_cmp(dcol, T.LE, 0.98, out_bitmap=bm, out_count=cnt))
report("K1 select_cmp DOUBLE col<=K", ms, N, 8 * N)

# the reference's DATE (8-byte DateLit: year, month, day) and CHAR(10) attributes: l_shipdate <= DATE, c_mktsegment = 'BUILDING'
dates = (torch.randint(1992, 1999, (N,), device=dev, generator=g, dtype=torch.int64) |
         (torch.randint(1, 13, (N,), device=dev, generator=g, dtype=torch.int64) << 32) |
         (torch.randint(1, 29, (N,), device=dev, generator=g, dtype=torch.int64) << 40))
ms = timed(lambda: capi.select_cmp(dates, T.LE, T.date_raw(1998, 9, 2), out_bitmap=bm, out_count=cnt, qtype=T.DATE))
report("K1 select_cmp DATE col<=K", ms, N, 8 * N)
del dates
nchar = N // 4
words = torch.tensor([list(w.ljust(10, b"\0")) for w in (b"AUTOMOBILE", b"BUILDING", b"FURNITURE", b"MACHINERY", b"HOUSEHOLD")],
                     dtype=torch.uint8, device=dev)
segment = words[torch.randint(0, 5, (nchar,), device=dev, generator=g)].contiguous()
ms = timed(lambda: capi.select_cmp_char(segment, T.EQ, b"BUILDING"))
report("K1 select_cmp_char CHAR(10) = 'BUILDING'", ms, nchar, 10 * nchar)
del segment

# compressed attribute: the same predicate on a 1-byte code stripe (dictionary / truncated column), and the decode
codes = torch.randint(0, 50, (N,), device=dev, generator=g, dtype=torch.uint8)
ms = timed(lambda: capi.select_codes(codes, T.CODE_LT, 24))
report("K1 select_codes 1-byte codes, code<K (stands for a DOUBLE column)", ms, N, N, "bytes of the code stripe; the DOUBLE column would be 8x")
dictionary = torch.arange(1, 51, device=dev, dtype=torch.float64)
ms = timed(lambda: capi.decode_codes(codes, dictionary, torch.float64))
report("decode 1-byte dictionary codes -> DOUBLE", ms, N, 9 * N)

# ---- K3 build / K5 gather ------------------------------------------------------------------------
nb = int(1_000_000 * max(scale, 0.1))
build = torch.randperm(nb, device=dev, generator=g, dtype=torch.int32)
table = capi.JoinTable(T.INT, nb)
ms = timed(lambda: (table.clear(), table.build(build)))
report("K3 join_build 1 M INT keys (clear + build)", ms, nb, 4 * nb)
probe = torch.randint(0, nb, (N,), device=dev, generator=g, dtype=torch.int32)
outs = (torch.empty(N, dtype=torch.int32, device=dev), torch.empty(N, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int64, device=dev))
ms = timed(lambda: table.probe(probe, capacity=N, out=outs))
report("K4 join_probe pairs (m=1.0)", ms, N, 12 * N)
dense = capi.JoinTable(T.INT, nb, key_range=(0, nb - 1))
ms = timed(lambda: (dense.clear(), dense.build(build)))
report("K3 dense join_build 1 M INT keys (clear + build)", ms, nb, 4 * nb)
ms = timed(lambda: dense.probe(probe, capacity=N, out=outs))
report("K4 dense join_probe pairs (m=1.0; exact min/max statistics)", ms, N, 12 * N)
ms = timed(lambda: dense.probe_count(probe))
report("K4 dense join_probe count only", ms, N, 4 * N)
payload = torch.rand(nb, device=dev, generator=g, dtype=torch.float64)
gout = torch.empty(N, dtype=torch.float64, device=dev)
ms = timed(lambda: capi.gather(payload, outs[1], out=gout))
report("K5 gather 8-B build payload by build_tid", ms, N, 16 * N, "random 8-B reads of an 8 MB column")

# ---- K12 LIP ----------------------------------------------------------------------------------------
lip = capi.LipFilter(T.LIP_BITVECTOR_EXACT, nb, 0)
lip.build(build)
ms = timed(lambda: lip.probe(probe))
report("K12 lip_probe exact filter", ms, N, 4 * N + N // 8)

# ---- K9 partition scatter (local half of the 8-GPU shuffle) -----------------------------------------
tids = torch.arange(N, dtype=torch.int32, device=dev)
ms = timed(lambda: capi.partition_scatter(probe, 8, [probe, tids]), reps=3)
report("K9 partition_scatter P=8, key + tid", ms, N, 4 * N * 2 + 8 * N * 2, "keys read twice (hist + scatter), 8 B/row moved")

# ---- aggregation family --------------------------------------------------------------------------------
na = 2 * N
cols = gen_q1_columns_gpu(na, dev, 4)
for label, create_env, update_env in (("AOT plan shape", None, None), ("run-time plan shape", "1", None),
                                      ("interpreter", "1", "1")):
    # QSX_AGG_NO_SPECIALIZE at creation: no AOT shape; still set at update: no run-time shape either
    if create_env:
        os.environ["QSX_AGG_NO_SPECIALIZE"] = create_env
    st = capi.AggState(q1_config())
    os.environ.pop("QSX_AGG_NO_SPECIALIZE", None)
    if update_env:
        os.environ["QSX_AGG_NO_SPECIALIZE"] = update_env
    ms = timed(lambda: st.update(cols, na), reps=3)
    os.environ.pop("QSX_AGG_NO_SPECIALIZE", None)
    report(f"K6 aggregate Q1 shape ({label})", ms, na, 34 * na)
k1 = torch.randint(0, 100, (na,), device=dev, generator=g, dtype=torch.int32)
k2 = torch.randint(0, 100, (na,), device=dev, generator=g, dtype=torch.int32)
val = torch.rand(na, device=dev, generator=g, dtype=torch.float64)
mincfg = T.make_agg_config(T.AGG_COMPACT_KEY, [(T.INT, None), (T.INT, None), (T.DOUBLE, None)], keys=[0, 1],
                           aggs=[(T.AGG_SUM, T.col(2)), (T.AGG_COUNT_STAR, None), (T.AGG_AVG, T.col(2))], est_groups=10_000)
st = capi.AggState(mincfg)
ms = timed(lambda: st.update([k1, k2, val], na), reps=3)
report("K6 aggregate 2 INT keys + DOUBLE, 10 k groups (plan shape)", ms, na, 16 * na)
gencfg = T.make_agg_config(T.AGG_GENERIC, [(T.INT, None), (T.INT, None), (T.DOUBLE, None)], keys=[0, 1],
                           aggs=[(T.AGG_SUM, T.col(2)), (T.AGG_COUNT_STAR, None)], est_groups=10_000)
st = capi.AggState(gencfg)
ms = timed(lambda: st.update([k1, k2, val], na), reps=3)
report("K8 aggregate GENERIC 2 INT keys, 10 k groups (group directory: key box; run-time plan shape)", ms, na, 16 * na)
st = capi.AggState(gencfg)
s1, s2 = k1 * 1_000_003, k2 * 7_919 - 11
ms = timed(lambda: st.update([s1, s2, val], na), reps=3)
del s1, s2
report("K8 aggregate GENERIC 2 INT keys spread over the INT range, 10 k groups (group directory: looked up)", ms, na, 16 * na)
os.environ["QSX_AGG_DIRECTORY"] = "0"
st = capi.AggState(gencfg)
ms = timed(lambda: st.update([k1, k2, val], na), reps=3)
os.environ.pop("QSX_AGG_DIRECTORY")
report("K8 the same without the group directory (K9 on the key code + per-piece tables: round 1)", ms, na, 16 * na)
# the reference's default group-by path at scale (PackedPayloadHashTable): 10^6 and 10^7 groups, random keys — two partition passes on
# hash digits, then 4096 pieces through LDS tables (csrc/agg_pieces.hpp); and the one-pass path it replaced
for many in (1_000_000, 10_000_000):
    big = torch.randint(0, many, (na,), device=dev, generator=g, dtype=torch.int32)
    bigcfg = T.make_agg_config(T.AGG_GENERIC, [(T.INT, None), (T.DOUBLE, None)], keys=[0], aggs=[(T.AGG_SUM, T.col(1)), (T.AGG_COUNT_STAR, None)],
                               est_groups=many)
    for env, label in ((None, "two partition passes + LDS pieces"), ("0", "one partition pass, then global atomics: round 5")):
        if env is not None:
            os.environ["QSX_AGG_TWO_LEVEL_MIN_GROUPS"] = env
        st = capi.AggState(bigcfg)

        def many_groups():
            st.clear()
            st.update([big, val], na)
        ms = timed(many_groups, reps=3)
        os.environ.pop("QSX_AGG_TWO_LEVEL_MIN_GROUPS", None)
        report(f"K8 aggregate GENERIC INT key, {many:,} groups, SUM + COUNT ({label})", ms, na, 12 * na, "state cleared inside the timed call")
        del st
    del big
widecfg = T.make_agg_config(T.AGG_GENERIC, [(T.INT, None), (T.LONG, None), (T.INT, None), (T.DOUBLE, None)], keys=[0, 1, 2],
                            aggs=[(T.AGG_SUM, T.col(3)), (T.AGG_COUNT_STAR, None)], est_groups=10_000)
st = capi.AggState(widecfg)
k_long, k_bit = k2.long() << 33, k1 & 1
ms = timed(lambda: st.update([k1, k_long, k_bit, val], na), reps=3)
report("K8 aggregate 16-byte key (INT, LONG, INT), 10 k groups (group directory: entries carry the key words)", ms, na, 24 * na)
del k_long, k_bit
# Q3 group-by shape: dense key (orders at SF100: 150 M keys for 600 M lineitems -> 4 rows per key, clustered)
ne = na // 4
okey = (torch.arange(na, device=dev, dtype=torch.int64) // 4).to(torch.int32)
price = torch.rand(na, device=dev, generator=g, dtype=torch.float64) * 1e5
disc = torch.randint(0, 11, (na,), device=dev, generator=g).double() / 100
cfcfg = T.make_agg_config(T.AGG_COLLISION_FREE, [(T.INT, None), (T.DOUBLE, None), (T.DOUBLE, None)], keys=[0],
                          instrs=[(T.EX_SUB, 0, T.const(0), T.col(2)), (T.EX_MUL, 1, T.col(1), T.temp(0))], consts=[1.0],
                          aggs=[(T.AGG_SUM, T.temp(1))], num_entries=ne)
st = capi.AggState(cfcfg)
ms = timed(lambda: st.update([okey, price, disc], na), reps=3)
report("K7 aggregate COLLISION_FREE clustered key, SUM(price*(1-disc))", ms, na, 20 * na, "random 8-B atomics on a dense array excluded")
