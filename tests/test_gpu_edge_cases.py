"""Edge cases through the C ABI: zero-length inputs for every entry point, sizes around the 64-row word
and 256/2048/4096-row tile boundaries, tuple-id range limits, invalid arguments reported as statuses."""
import numpy as np
import pytest
import torch

from quickstep_amd import types as T
from helpers import bitmap_np, to_dev

pytestmark = pytest.mark.gpu


def test_zero_length_inputs_everywhere(capi, dev):
    e32 = torch.empty(0, dtype=torch.int32, device=dev)
    e64f = torch.empty(0, dtype=torch.float64, device=dev)
    bm, cnt = capi.select_cmp(e32, T.LT, 5)
    assert int(cnt.item()) == 0
    bm, cnt = capi.select_cmp_columns(e32, e32, T.EQ)
    assert int(cnt.item()) == 0
    cols, k = capi.compact_gather([e32], bm, 0)
    assert int(k.item()) == 0
    tids, k = capi.bitmap_to_tids(bm, 0)
    assert int(k.item()) == 0
    assert capi.gather(to_dev(np.arange(4, dtype=np.int32), dev), e32).numel() == 0
    keys, exact = capi.join_key_pack([e32, e32])
    assert keys.numel() == 0 and exact
    scattered, offsets = capi.partition_scatter(e32, 8, [e32])
    assert offsets.cpu().tolist() == [0] * 9
    f = capi.LipFilter(T.LIP_BITVECTOR_EXACT, 100, 0)
    f.build(e32)
    bm, cnt = f.probe(e32)
    assert int(cnt.item()) == 0
    for key_range in (None, (0, 9)):
        t = capi.JoinTable(T.INT, 0, key_range=key_range)
        t.build(e32)
        assert t.size() == 0
        assert int(t.probe_count(e32).item()) == 0
    # aggregation: an update with no rows leaves the freshly initialised state (COUNT 0, SUM/MIN NULL)
    cfg = T.make_agg_config(T.AGG_SINGLE_STATE, [(T.DOUBLE, None)], aggs=[(T.AGG_SUM, T.col(0)), (T.AGG_MIN, T.col(0)), (T.AGG_COUNT_STAR, None)])
    st = capi.AggState(cfg)
    st.update([e64f], 0)
    keys, vals, nulls, groups = st.finalize(dev)
    assert int(groups.item()) == 1 and int(vals[2][0].item()) == 0 and int(nulls[0][0].item()) == 1 and int(nulls[1][0].item()) == 1
    cfg = T.make_agg_config(T.AGG_COMPACT_KEY, [(T.INT, None), (T.DOUBLE, None)], keys=[0], aggs=[(T.AGG_SUM, T.col(1))])
    st = capi.AggState(cfg)
    st.update([e32, e64f], 0)
    assert st.num_groups() <= 1                       # an upper bound (the reserved sentinel slot counts)
    assert int(st.finalize(dev)[3].item()) == 0


@pytest.mark.parametrize("n", [1, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 4095, 4096, 4097, 12_289])
def test_sizes_around_word_and_tile_boundaries(capi, oracle, dev, n):
    rng = np.random.default_rng(n)
    key = rng.integers(0, 37, size=n).astype(np.int32)
    val = rng.integers(-50, 50, size=n).astype(np.int64)
    dk, dv = to_dev(key, dev), to_dev(val, dev)
    # select + compaction
    bm, cnt = capi.select_cmp(dk, T.GE, 11)
    ref = oracle.select_cmp(key, T.GE, 11)
    assert np.array_equal(bitmap_np(bm), ref)
    (out,), k = capi.compact_gather([dv], bm, n)
    assert np.array_equal(out[:int(k.item())].cpu().numpy(), val[key >= 11])
    # join (both flavours): every probe row matches the rows with its key
    for key_range in (None, (0, 36)):
        t = capi.JoinTable(T.INT, n, key_range=key_range)
        t.build(dk)
        total = int(t.probe_count(dk).item())
        assert total == int((np.bincount(key, minlength=37) ** 2).sum())
        p, b, c = t.probe(dk, capacity=total)
        assert int(c.item()) == total
        assert np.array_equal(key[p[:total].cpu().numpy()], key[b[:total].cpu().numpy()])
    # partition scatter: stable, 5 partitions (not a power of two)
    (sk, sv), offsets = capi.partition_scatter(dk, 5, [dk, dv])
    pid = key % 5
    want = np.concatenate([val[pid == p] for p in range(5)])
    assert np.array_equal(sv.cpu().numpy(), want)
    assert offsets.cpu().tolist() == [0] + np.cumsum(np.bincount(pid, minlength=5)).tolist()
    # aggregation
    cfg = T.make_agg_config(T.AGG_COMPACT_KEY, [(T.INT, None), (T.LONG, None)], keys=[0],
                            aggs=[(T.AGG_SUM, T.col(1)), (T.AGG_MAX, T.col(1)), (T.AGG_COUNT_STAR, None)], est_groups=40)
    st = capi.AggState(cfg)
    st.update([dk, dv], n)
    keys, vals, nulls, groups = st.finalize(dev)
    g = int(groups.item())
    order = np.argsort(keys[0][:g].cpu().numpy())
    present = np.unique(key)
    assert np.array_equal(keys[0][:g].cpu().numpy()[order], present)
    assert np.array_equal(vals[0][:g].cpu().numpy()[order], np.array([val[key == k].sum() for k in present]))
    assert np.array_equal(vals[1][:g].cpu().numpy()[order], np.array([val[key == k].max() for k in present]))


def test_tuple_id_range_and_argument_errors(capi, dev):
    keys = to_dev(np.arange(10, dtype=np.int32), dev)
    t = capi.JoinTable(T.INT, 10)
    with pytest.raises(capi.QsxError):
        t.build(keys, base_tid=2**31 - 5)              # base_tid + n exceeds the 32-bit tuple reference
    with pytest.raises(capi.QsxError):
        t.build(keys, base_tid=-1)
    t.build(keys, base_tid=2**31 - 11)                 # the largest legal base
    p, b, c = t.probe(keys, capacity=10, probe_base_tid=2**31 - 11)
    assert int(c.item()) == 10
    assert int(b[:10].max().item()) == 2**31 - 2 and int(p[:10].max().item()) == 2**31 - 2
    with pytest.raises(capi.QsxError):
        capi.JoinTable(7, 10)                          # unknown key type
    with pytest.raises(capi.QsxError):
        capi.LipFilter(T.LIP_BITVECTOR_EXACT, 0, 0)    # empty filter
    with pytest.raises(capi.QsxError):
        capi.partition_scatter(keys, 65, [keys])       # more partitions than a wave has lanes
    with pytest.raises(capi.QsxError):
        capi.AggState(T.make_agg_config(T.AGG_COMPACT_KEY, [(T.LONG, None)] * 4, keys=[0, 1, 2, 3], aggs=[]))   # 32-byte key: four words
    capi.AggState(T.make_agg_config(T.AGG_COMPACT_KEY, [(T.LONG, None), (T.LONG, None)], keys=[0, 1], aggs=[])).close()   # 16 bytes: a wide key


def test_projecting_probe_argument_errors_and_empty_sides(capi, dev):
    """qsx_join_probe_project_blocks: what it refuses (column counts, widths, missing stripes, descending segment starts)
    and what it accepts without touching memory that is not there (no probe rows, a build side without tuples, capacity 0:
    the count only)."""
    import ctypes as C
    keys = torch.arange(1000, dtype=torch.int32, device=dev)
    t = capi.JoinTable(T.INT, 1000, key_range=(0, 999))
    t.build(keys)
    probe = torch.randint(0, 2000, (5000,), dtype=torch.int32, device=dev)
    expected = int((probe < 1000).sum().item())
    # capacity 0: the count, nothing written
    outs, cnt = t.probe_project_blocks([probe], [[probe]], [[keys]], capacity=0)
    assert int(cnt.item()) == expected
    # a smaller capacity than the result: the full count, no write past the capacity
    outs, cnt = t.probe_project_blocks([probe], [[probe]], [[keys]], capacity=7)
    assert int(cnt.item()) == expected and outs[0].numel() == 7 and bool((outs[0] == outs[1]).all())
    # no probe rows at all
    none = torch.empty(0, dtype=torch.int32, device=dev)
    outs, cnt = t.probe_project_blocks([none, none], [[none, none]], [[keys]])
    assert int(cnt.item()) == 0
    # an empty table, its build relation without tuples
    empty = capi.JoinTable(T.INT, 10, key_range=(0, 9))
    outs, cnt = empty.probe_project_blocks([probe], [[probe]], [[none]])
    assert int(cnt.item()) == 0
    # refused descriptors
    def call(proj):
        count = torch.zeros(1, dtype=torch.int64, device=dev)
        rows = (C.c_int64 * 1)(probe.numel())
        kptr = (C.c_void_p * 1)(probe.data_ptr())
        return capi.lib.qsx_join_probe_project_blocks(t._h, 1, rows, kptr, None, C.byref(proj), 100, C.c_void_p(count.data_ptr()), None)
    out = torch.empty(100, dtype=torch.int64, device=dev)
    def descriptor(num_columns=1, width=4, on_build=0, segments=1, first=(0,)):
        p = T.JoinProjection()
        p.num_columns = num_columns
        p.width[0] = width
        p.on_build[0] = on_build
        stripes = (C.c_void_p * 1)(probe.data_ptr())
        bstripes = (C.c_void_p * max(segments, 1))(*([keys.data_ptr()] * max(segments, 1)))
        firsts = (C.c_int64 * max(segments, 1))(*first)
        outp = (C.c_void_p * 1)(out.data_ptr())
        p.probe_stripes = C.cast(stripes, C.POINTER(C.c_void_p))
        p.num_build_segments = segments
        p.build_first_tids = C.cast(firsts, C.POINTER(C.c_int64))
        p.build_stripes = C.cast(bstripes, C.POINTER(C.c_void_p))
        p.out_columns = C.cast(outp, C.POINTER(C.c_void_p))
        p._keep = (stripes, bstripes, firsts, outp)
        return p
    assert call(descriptor()) == 0
    assert call(descriptor(num_columns=0)) == T.ERR_INVALID_ARGUMENT
    assert call(descriptor(num_columns=T.MAX_PROJECTED + 1)) == T.ERR_INVALID_ARGUMENT
    assert call(descriptor(width=3)) == T.ERR_INVALID_ARGUMENT
    assert call(descriptor(width=16)) == T.ERR_INVALID_ARGUMENT
    assert call(descriptor(on_build=1, segments=2, first=(500, 0))) == T.ERR_INVALID_ARGUMENT
    p = descriptor()
    p.out_columns = None
    assert call(p) == T.ERR_INVALID_ARGUMENT
    p = descriptor()
    p.probe_stripes = None
    assert call(p) == T.ERR_INVALID_ARGUMENT



def test_calls_follow_the_calling_threads_device(capi, dev):
    """Nothing in the library is bound to device 0: a call works on the device current in its thread (include/qsx.h,
    qsx_current_device / qsx_set_current_device).  A thread the engine creates starts on device 0 whatever the creating
    thread had selected — it selects explicitly, which is what the host layer's Worker threads do."""
    import threading
    assert capi.current_device() == torch.cuda.current_device()
    count = capi.device_count()
    for bad in (-1, count):
        with pytest.raises(capi.QsxError) as e:
            capi.set_current_device(bad)
        assert e.value.status == T.ERR_INVALID_ARGUMENT
    seen = []
    def worker():
        capi.set_current_device(count - 1)
        seen.append(capi.current_device())
    t = threading.Thread(target=worker)
    t.start()
    t.join()
    assert seen == [count - 1]
    assert capi.current_device() == torch.cuda.current_device()


def test_idle_allocations_are_reused_on_their_own_device_only(capi, dev):
    """the idle-allocation cache (a destroyed table's / state's memory kept for the next create of the same
    size) is keyed by (device, size).  One device: destroy -> create of the same size hands the same memory back.  With a
    second device in the box: a create issued from a thread on device 1 must NOT receive device 0's memory."""
    cfg = T.make_agg_config(T.AGG_COLLISION_FREE, [(T.INT, None), (T.DOUBLE, None)], keys=[0], aggs=[(T.AGG_SUM, T.col(1))], num_entries=777_777)
    st = capi.AggState(cfg)
    k0 = torch.arange(0, 1000, device=dev, dtype=torch.int32)
    st.update([k0, torch.ones(1000, device=dev, dtype=torch.float64)], 1000)
    assert int(st.finalize(dev)[3].item()) == 1000
    st.close()
    st2 = capi.AggState(cfg)                   # same size, same device: the idle allocation comes back, cleared
    assert int(st2.finalize(dev)[3].item()) == 0
    st2.close()
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU in this box: the cross-device half of this test needs two")
    import threading
    seen = {}

    def on_device_1():
        capi.set_current_device(1)
        d1 = torch.device("cuda", 1)
        other = capi.AggState(cfg)             # would be handed device 0's idle image if the cache were keyed by size alone
        k = torch.randint(0, 777_777, (100_000,), device=d1, dtype=torch.int32)
        v = torch.ones(100_000, device=d1, dtype=torch.float64)
        other.update([k, v], k.numel())
        _, vals1, _, g1 = other.finalize(d1)
        seen["groups"] = int(g1.item())
        seen["sum"] = float(vals1[0][: seen["groups"]].sum().item())
        seen["want_groups"] = int(torch.unique(k).numel())
        other.close()

    t = threading.Thread(target=on_device_1)
    t.start()
    t.join()
    assert seen["groups"] == seen["want_groups"] and seen["sum"] == 100_000.0


def test_copy_segments_equals_a_copy_per_segment(capi, dev):
    """qsx_copy_segments: many device-to-device copies in one launch — 16-byte aligned pieces (16 bytes a lane), pieces at odd
    addresses and of odd lengths (byte by byte), empty pieces, more pieces than one grid dimension takes."""
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(9)
    pool = torch.randint(0, 256, (9_000_000,), device=dev, generator=g, dtype=torch.uint8)
    out = torch.zeros_like(pool)
    cuts = [0, 16, 16, 4096 * 16 + 32, 1_000_003, 1_000_003 + 7, 3_000_000, 3_000_016, 8_999_999, 9_000_000]
    srcs = [pool[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    dsts = [out[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    capi.copy_segments(srcs, dsts)
    assert torch.equal(out, pool)
    # 70 000 small pieces, shuffled destinations
    n, piece = 70_000, 48
    src = pool[: n * piece].view(n, piece)
    perm = torch.randperm(n, device=dev, generator=g)
    dst = torch.zeros(n, piece, dtype=torch.uint8, device=dev)
    p = perm.cpu().tolist()
    capi.copy_segments([src[i] for i in range(n)], [dst[p[i]] for i in range(n)])
    assert torch.equal(dst[perm], src)
    capi.copy_segments([], [])
    # pieces whose addresses and lengths are multiples of 8, 4 and 2 bytes only (a stripe of 349 525 INTs: 4-byte units)
    out.zero_()
    pieces = [(8, 8 + 16 * 1000 + 8), (40_004, 40_004 + 4 * 349_525), (2_000_002, 2_000_002 + 2 * 77_777), (3_000_001, 3_000_001 + 12_345)]
    capi.copy_segments([pool[a:b] for a, b in pieces], [out[a + 16:b + 16] for a, b in pieces])      # (the same residues: + 16)
    for a, b in pieces:
        assert torch.equal(out[a + 16:b + 16], pool[a:b])
    assert int(out.count_nonzero().item()) <= sum(b - a for a, b in pieces)


def test_partition_scatter_blocks_argument_errors_and_empty_runs(capi, dev):
    """qsx_partition_scatter_blocks: no blocks / only empty blocks leave P + 1 zero offsets; a block with rows and no key stripe, a
    negative row count, a column width K9 does not move, more than 64 partitions and a workspace that is too small are refused."""
    import ctypes as C
    import torch
    from quickstep_amd import types as T
    lib = capi.lib
    P = 4
    offs = torch.full((P + 1,), 7, dtype=torch.int64, device=dev)
    ws = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
    keys = torch.arange(5000, dtype=torch.int32, device=dev)
    out = torch.empty_like(keys)

    def call(rows, key_ptrs, col_ptrs, widths=(4,), parts=P, ws_bytes=ws.numel(), key_type=T.INT):
        nb = len(rows)
        return lib.qsx_partition_scatter_blocks(key_type, nb, (C.c_int64 * max(nb, 1))(*rows), (C.c_void_p * max(nb, 1))(*key_ptrs), parts,
                                                len(widths), (C.c_void_p * max(len(col_ptrs), 1))(*col_ptrs), (C.c_int32 * len(widths))(*widths),
                                                (C.c_void_p * len(widths))(*([out.data_ptr()] * len(widths))), C.c_void_p(offs.data_ptr()),
                                                C.c_void_p(ws.data_ptr()), ws_bytes, None)

    assert call([], [], []) == T.OK
    torch.cuda.synchronize()
    assert offs.tolist() == [0] * (P + 1)
    offs.fill_(7)
    assert call([0, 0], [None, None], [None, None]) == T.OK                       # empty blocks: their pointers are not looked at
    torch.cuda.synchronize()
    assert offs.tolist() == [0] * (P + 1)
    k = keys.data_ptr()
    assert call([5000], [None], [k]) == T.ERR_INVALID_ARGUMENT
    assert call([5000], [k], [None]) == T.ERR_INVALID_ARGUMENT
    assert call([-1], [k], [k]) == T.ERR_INVALID_ARGUMENT
    assert call([5000], [k], [k], widths=(3,)) == T.ERR_UNSUPPORTED
    assert call([5000], [k], [k], parts=65) == T.ERR_UNSUPPORTED
    assert call([5000], [k], [k], parts=0) == T.ERR_INVALID_ARGUMENT
    assert call([5000], [k], [k], key_type=T.DOUBLE) == T.ERR_UNSUPPORTED
    assert call([5000], [k], [k], ws_bytes=8) == T.ERR_CAPACITY
    assert lib.qsx_partition_blocks_workspace_bytes(5000, 1, P) <= ws.numel()
    assert call([2000, 0, 3000], [k, None, k + 4 * 2000], [k, None, k + 4 * 2000]) == T.OK   # the same rows as two blocks around an empty one
    torch.cuda.synchronize()
    (want,), want_offs = capi.partition_scatter(keys, P, [keys])
    assert torch.equal(out, want) and torch.equal(offs, want_offs)


def test_coded_block_forms_argument_errors_and_plain_equivalence(capi, dev):
    """qsx_*_blocks_coded: a coding with a width other than 0 / 1 / 2 / 4, or a dictionary for a block that holds values, is an
    invalid argument; a NULL coding and a coding of all zeros ARE the plain forms; empty runs do nothing."""
    import ctypes as C
    import torch
    from quickstep_amd import types as T
    lib = capi.lib
    keys = torch.arange(1000, dtype=torch.int32, device=dev)
    table = capi.JoinTable(T.INT, 1000, key_range=(0, 999))
    table.build_blocks([keys[:400], keys[400:]], [0, 400], coding=[(0, None), (0, None)])       # all zeros: the plain form
    assert table.size() == 1000
    probe = torch.randint(0, 2000, (5000,), device=dev, dtype=torch.int32)
    want = int(table.probe_count_blocks([probe]).item())
    assert int(table.probe_count_blocks([probe], coding=[(0, None)]).item()) == want
    with pytest.raises(capi.QsxError) as e:
        table.probe_count_blocks([probe], coding=[(3, None)])
    assert e.value.status == T.ERR_INVALID_ARGUMENT
    d = torch.zeros(4, dtype=torch.int32, device=dev)
    with pytest.raises(capi.QsxError) as e:
        table.probe_count_blocks([probe], coding=[(0, d)])                                          # a dictionary without codes
    assert e.value.status == T.ERR_INVALID_ARGUMENT
    with pytest.raises(capi.QsxError) as e:
        table.build_blocks([keys], [0], coding=[(8, None)])
    assert e.value.status == T.ERR_INVALID_ARGUMENT
    # NULL coding struct / NULL width array through the raw entry point
    rows = (C.c_int64 * 1)(probe.numel())
    kptr = (C.c_void_p * 1)(probe.data_ptr())
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    assert lib.qsx_join_probe_count_blocks_coded(table._h, 1, rows, kptr, None, None, C.c_void_p(count.data_ptr()), None) == 0
    assert int(count.item()) == want
    empty = capi.KeyCoding(None, None)
    assert lib.qsx_join_probe_count_blocks_coded(table._h, 1, rows, kptr, C.byref(empty), None, C.c_void_p(count.data_ptr()), None) == 0
    assert int(count.item()) == want
    assert int(table.probe_count_blocks([], coding=[]).item()) == 0
    lip = capi.LipFilter(T.LIP_BITVECTOR_EXACT, 2000, min_value=0)
    with pytest.raises(capi.QsxError) as e:
        lip.build_blocks([keys], coding=[(5, None)], key_type=T.INT)
    assert e.value.status == T.ERR_INVALID_ARGUMENT
    with pytest.raises(capi.QsxError) as e:
        capi.join_key_pack_blocks_coded([[keys, keys]], [[(0, None), (7, None)]], [T.INT, T.INT])
    assert e.value.status == T.ERR_INVALID_ARGUMENT
