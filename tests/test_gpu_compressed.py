"""GPU parity for compressed attributes: code-stripe scans and decode through the C ABI
against the oracle's restatement of CompressedColumnStoreTupleStorageSubBlock / CompressedStoreUtil."""
import os

import numpy as np
import pytest
import torch

from quickstep_amd import types as T
from helpers import bitmap_dev, bitmap_np, to_dev
from test_oracle_pins import _compressed_cases

pytestmark = pytest.mark.gpu

_TORCH_CODES = {1: torch.uint8, 2: torch.int16, 4: torch.int32}   # same bytes as the unsigned codes


def codes_dev(col, dev):
    return torch.from_numpy(col.codes.view({1: np.uint8, 2: np.int16, 4: np.int32}[col.code_width]).copy()).to(dev)


@pytest.mark.parametrize("n", [1, 64, 4097, 300_001])
def test_code_stripe_scans_are_bit_exact(capi, oracle, dev, n):
    rng = np.random.default_rng(n)
    f = oracle.bitmap_from_bools(rng.random(n) < 0.6)
    for width, np_t in ((1, np.uint8), (2, np.uint16), (4, np.uint32)):
        hi = {1: 256, 2: 65536, 4: 2**32}[width]
        codes = rng.integers(0, hi, size=n, dtype=np.uint64).astype(np_t)
        d = torch.from_numpy(codes.view({1: np.uint8, 2: np.int16, 4: np.int32}[width]).copy()).to(dev)
        first = int(np.sort(codes)[n // 3])
        second = int(np.sort(codes)[2 * n // 3]) + 1
        for op, a, b in ((T.CODE_EQ, first, 0), (T.CODE_NE, first, 0), (T.CODE_LT, first, 0), (T.CODE_GE, first, 0),
                         (T.CODE_RANGE, first, second), (T.CODE_GE, 0, 0), (T.CODE_LT, 0, 0), (T.CODE_GE, hi - 1, 0),
                         (T.CODE_RANGE, second, first)):
            for filt in (None, f):
                bm, cnt = capi.select_codes(d, op, a, b, filter_bitmap=None if filt is None else bitmap_dev(filt, dev))
                want = oracle.select_codes(codes, op, a, b, filt)
                assert np.array_equal(bitmap_np(bm), want), (width, op)
                assert int(cnt.item()) == oracle.bitmap_count(want, n)


@pytest.mark.parametrize("n", [1, 65, 5000, 400_003])
def test_sorted_code_stripe_by_binary_search(capi, oracle, dev, n):
    """The sort column of a compressed column store (lineitem / orders / partsupp in the reference's TPC-H DDL): ascending
    codes, every code comparison answered by two searches — must equal the scan of the same stripe."""
    rng = np.random.default_rng(n + 3)
    f = oracle.bitmap_from_bools(rng.random(n) < 0.5)
    for width, np_t, hi in ((1, np.uint8, 256), (2, np.uint16, 65536), (4, np.uint32, 2**32)):
        codes = np.sort(rng.integers(0, min(hi, max(4, n // 3)), size=n, dtype=np.uint64)).astype(np_t)   # runs of equal codes
        if n > 100 and width == 4:
            codes[-3:] = hi - 1                                                                            # the largest code
        d = torch.from_numpy(codes.view({1: np.uint8, 2: np.int16, 4: np.int32}[width]).copy()).to(dev)
        picks = sorted({int(codes[0]), int(codes[n // 2]), int(codes[-1]), 0, min(hi - 1, int(codes[-1]) + 1)})
        for a in picks:
            for op, b in ((T.CODE_EQ, 0), (T.CODE_NE, 0), (T.CODE_LT, 0), (T.CODE_GE, 0), (T.CODE_RANGE, min(hi - 1, a + 7)),
                          (T.CODE_RANGE, max(a - 1, 0))):
                for filt in (None, f):
                    bm, cnt = capi.select_codes_sorted(d, op, a, b, filter_bitmap=None if filt is None else bitmap_dev(filt, dev))
                    want = oracle.select_codes(codes, op, a, b, filt)
                    assert np.array_equal(bitmap_np(bm), want), (n, width, op, a, b)
                    assert int(cnt.item()) == oracle.bitmap_count(want, n)


def test_compressed_columns_select_and_decode(capi, oracle, dev):
    """getMatchesForPredicate on compressed attributes: the caller's transform (oracle restatement here, the C++
    host layer in the product) + qsx_select_codes == the comparison on the uncompressed values through qsx_select_cmp;
    qsx_decode_codes gives the values back for operators that need them."""
    for name, values, kind, width in _compressed_cases():
        col = oracle.CompressedColumn(values)
        if kind == 0:
            continue
        d_codes = codes_dev(col, dev)
        d_values = to_dev(values, dev)
        d_dict = None if col.dictionary is None else to_dev(col.dictionary, dev)
        decoded = capi.decode_codes(d_codes, d_dict, d_values.dtype)
        assert torch.equal(decoded, d_values), name
        lits = [values.min(), values.max(), np.sort(values)[values.size // 2], values.max() + 1]
        for lit in lits:
            lit = values.dtype.type(lit)
            for op in range(6):
                want, want_cnt = capi.select_cmp(d_values, op, lit.item())
                pred = col.transform(op, lit)
                if pred.result == oracle.PRED_NONE:
                    assert int(want_cnt.item()) == 0, (name, op, lit)
                elif pred.result == oracle.PRED_ALL:
                    assert int(want_cnt.item()) == values.size, (name, op, lit)
                else:
                    got, cnt = capi.select_codes(d_codes, pred.comp, pred.first, pred.second)
                    assert torch.equal(got, want) and int(cnt.item()) == int(want_cnt.item()), (name, op, lit)


@pytest.mark.parametrize("code_dtype", [np.uint8, np.uint16, np.uint32])
@pytest.mark.parametrize("value_dtype", [np.int32, np.int64, np.float32, np.float64])
def test_decode_codes_every_width_length_and_alignment(capi, dev, code_dtype, value_dtype):
    """qsx_decode_codes: dictionary lookup and zero-extension for 1 / 2 / 4-byte codes into 4 / 8-byte values — the kernel that
    writes 16 bytes per lane (two or four values from one read of their codes) for the whole groups of an aligned stripe, the
    one-value kernel for the rows behind the last group and for stripes that start at odd addresses.  Lengths on both sides of a
    group, a workgroup's share and the grid; equal to numpy's `dictionary[codes]` / `codes.astype`."""
    rng = np.random.default_rng(5)
    entries = min(int(np.iinfo(code_dtype).max) + 1, 40_000)
    if np.issubdtype(value_dtype, np.integer):
        dictionary = np.sort(rng.choice(np.arange(-10**6, 10**6), size=entries, replace=False)).astype(value_dtype)
    else:
        dictionary = np.sort(rng.standard_normal(entries)).astype(value_dtype)
    d_dict = to_dev(dictionary, dev)
    for n in (1, 2, 3, 4, 5, 1023, 4096, 4099, 1_048_577, 3_000_001):
        for skew in (0, 1, 3):                         # rows by which the stripes start behind an aligned address
            pool = rng.integers(0, entries, size=n + skew).astype(code_dtype)
            d_pool = to_dev(pool.view({1: np.uint8, 2: np.int16, 4: np.int32}[pool.itemsize]), dev)   # (the bits are what travels)
            out_pool = torch.zeros(n + skew + 4, dtype=d_dict.dtype, device=dev)
            got = capi.decode_codes(d_pool[skew:], d_dict, out_pool.dtype, out=out_pool[skew:skew + n])
            assert np.array_equal(got.cpu().numpy(), dictionary[pool[skew:]]), (n, skew)
            assert not out_pool[:skew].any() and not out_pool[skew + n:].any(), (n, skew)   # nothing written outside the stripe
            if np.issubdtype(value_dtype, np.integer):   # truncated values: zero-extension
                got = capi.decode_codes(d_pool[skew:], None, out_pool.dtype)
                assert np.array_equal(got.cpu().numpy(), pool[skew:].astype(value_dtype)), (n, skew)


def test_q1_style_aggregation_over_decoded_dictionary_columns(capi, oracle, dev):
    """lineitem in the reference's TPC-H DDL is a compressed column store (benchmarks/tpch/create.sql:69-121): l_discount,
    l_tax, l_quantity are dictionary-coded.  Predicate on the l_quantity codes, decode the arguments, aggregate."""
    rng = np.random.default_rng(3)
    n = 500_000
    qty = rng.integers(1, 51, size=n).astype(np.float64)
    disc = rng.integers(0, 11, size=n) / 100.0
    price = np.round(rng.uniform(900, 105000, size=n), 2)
    cq, cd = oracle.CompressedColumn(qty), oracle.CompressedColumn(disc)
    assert cq.kind == 2 and cd.kind == 2 and cq.code_width == 1
    pred = cq.transform(T.LT, 24.0)
    bm, cnt = capi.select_codes(codes_dev(cq, dev), pred.comp, pred.first, pred.second)
    assert int(cnt.item()) == int((qty < 24).sum())
    d_disc = capi.decode_codes(codes_dev(cd, dev), to_dev(cd.dictionary, dev), torch.float64)
    cfg = T.make_agg_config(T.AGG_SINGLE_STATE, [(T.DOUBLE, None), (T.DOUBLE, None)],
                            instrs=[(T.EX_MUL, 0, T.col(0), T.col(1))], aggs=[(T.AGG_SUM, T.temp(0)), (T.AGG_COUNT_STAR, None)])
    st = capi.AggState(cfg)
    st.update([to_dev(price, dev), d_disc], n, filter_bitmap=bm)
    keys, vals, nulls, groups = st.finalize(dev)
    sel = qty < 24
    assert int(vals[1][0].item()) == int(sel.sum())
    assert np.isclose(vals[0][0].item(), (price[sel] * disc[sel]).sum(), rtol=1e-9)


def _q1_coded_inputs(oracle, rng, n):
    """Q1's lineitem columns as CompressedBlockBuilder would store them: quantity / discount / tax dictionary-coded in
    one byte, the keys plain CHAR(1), extendedprice left as DOUBLE (too many distinct values), plus a truncated INT."""
    combo = rng.choice(4, size=n, p=[0.2466, 0.0065, 0.5005, 0.2464])
    k1 = np.frombuffer(b"ANNR", dtype=np.uint8)[combo]
    k2 = np.frombuffer(b"FFOF", dtype=np.uint8)[combo]
    qty = rng.integers(1, 51, size=n).astype(np.float64)
    price = np.round(rng.uniform(900, 105000, size=n), 2)
    disc = rng.integers(0, 11, size=n) / 100.0
    tax = rng.integers(0, 9, size=n) / 100.0
    lineno = rng.integers(1, 8, size=n).astype(np.int32)            # truncation-compressed: value = code
    cols = [k1, k2, qty, price, disc, tax, lineno]
    comp = [None, None] + [oracle.CompressedColumn(c) for c in (qty, price, disc, tax, lineno)]
    return cols, comp


@pytest.mark.parametrize("jit", [False, True])
def test_q1_over_compressed_attributes_matches_oracle(capi, oracle, dev, jit, monkeypatch):
    """qsx_agg_update_coded: the aggregation reads the code stripes (13 instead of 34 bytes per row for Q1) and decodes
    while staging; results equal the oracle aggregating the decoded columns (and the plain-column GPU path) —
    interpreter and run-time plan shape."""
    monkeypatch.setenv("QSX_AGG_JIT_MIN_ROWS", "0" if jit else str(1 << 60))
    rng = np.random.default_rng(77)
    n = 400_003
    cols, comp = _q1_coded_inputs(oracle, rng, n)
    widths = [0 if c is None or c.kind == 0 else c.code_width for c in comp]
    assert widths[2] == widths[4] == widths[5] == 1 and widths[3] == 0 and widths[6] == 1
    assert comp[2].dictionary is not None and comp[6].dictionary is None          # dictionary vs truncation
    layout = [(T.CHAR, 1), (T.CHAR, 1), (T.DOUBLE, None), (T.DOUBLE, None), (T.DOUBLE, None), (T.DOUBLE, None), (T.INT, None)]
    kw = dict(keys=[0, 1],
              instrs=[(T.EX_SUB, 0, T.const(0), T.col(4)), (T.EX_MUL, 1, T.col(3), T.temp(0)),
                      (T.EX_ADD, 2, T.const(0), T.col(5)), (T.EX_MUL, 3, T.temp(1), T.temp(2))],
              consts=[1.0],
              aggs=[(T.AGG_SUM, T.col(2)), (T.AGG_SUM, T.col(3)), (T.AGG_SUM, T.temp(1)), (T.AGG_SUM, T.temp(3)),
                    (T.AGG_AVG, T.col(4)), (T.AGG_COUNT_STAR, None), (T.AGG_SUM, T.col(6)), (T.AGG_MAX, T.col(4))],
              pred=[(6, T.LT, 7)], est_groups=6)
    cfg = T.make_agg_config(T.AGG_COMPACT_KEY, layout, code_widths=widths, **kw)
    plain_cfg = T.make_agg_config(T.AGG_COMPACT_KEY, layout, **kw)
    code_cols = [cols[i] if widths[i] == 0 else comp[i].codes for i in range(len(cols))]
    dicts = [None if widths[i] == 0 else comp[i].dictionary for i in range(len(cols))]
    st = capi.AggState(cfg)
    filt = oracle.bitmap_from_bools(rng.random(n) < 0.8)
    for lo, hi in ((0, 150_000), (150_000, n)):                    # two "blocks"; the second one under a filter
        f = None if lo == 0 else oracle.bitmap_from_bools(oracle.bools_from_bitmap(filt, n)[lo:hi])
        st.update_coded([to_dev(np.ascontiguousarray(c[lo:hi]), dev) for c in code_cols],
                        [None if d is None else to_dev(d, dev) for d in dicts], hi - lo,
                        filter_bitmap=None if f is None else bitmap_dev(f, dev))
    o = oracle.AggState(cfg)
    o.update_coded([np.ascontiguousarray(c[:150_000]) for c in code_cols], dicts, 150_000)
    o.update_coded([np.ascontiguousarray(c[150_000:]) for c in code_cols], dicts, n - 150_000,
                   filter_bitmap=oracle.bitmap_from_bools(oracle.bools_from_bitmap(filt, n)[150_000:]))
    from test_gpu_agg import assert_same_groups, finalize_np
    assert_same_groups(finalize_np(st, dev), o.finalize())
    # the decoded columns through the plain entry point give the same groups
    o2 = oracle.AggState(plain_cfg)
    o2.update([c[:150_000] for c in cols])
    o2.update([c[150_000:] for c in cols], filter_bitmap=oracle.bitmap_from_bools(oracle.bools_from_bitmap(filt, n)[150_000:]))
    assert_same_groups(o.finalize(), o2.finalize())
    with pytest.raises(capi.QsxError):                               # a coded state needs the coded entry point
        st.update([to_dev(c[:10], dev) for c in code_cols], 10)


@pytest.mark.parametrize("kernel", ["interpreter", "plan_shape", "plan_shape_unsized_dictionaries", "plan_shape_register_groups"])
def test_group_by_compressed_keys_matches_oracle(capi, oracle, dev, kernel, monkeypatch):
    """Hash-strategy states whose GROUP BY attributes themselves arrive as codes (a dictionary-coded INT, a truncated LONG)
    next to coded arguments: the plan shapes keep only the code stripes in the tile and decode a thread's rows into
    registers (agg_hash_update.hpp DecodedRows) — key packing, the predicate and the aggregates' arguments all read them
    there; the interpreter decodes into LDS slots.  Also with per-wave register accumulators (QSX_AGG_REG_GROUPS=1), a
    handful of groups and then more groups than the registers keep.  The dictionaries' sizes travel with the call
    (qsx_agg_update_coded_sized): those of up to 64 entries are decoded from LDS (k1, quantity), the 150-entry one and — in
    the unsized variant, plain qsx_agg_update_coded — all of them through memory."""
    monkeypatch.setenv("QSX_AGG_JIT_MIN_ROWS", str(1 << 60) if kernel == "interpreter" else "0")
    monkeypatch.setenv("QSX_AGG_REG_GROUPS", "1" if kernel == "plan_shape_register_groups" else "0")
    rng = np.random.default_rng(91)
    n = 300_007
    from test_gpu_agg import assert_same_groups, finalize_np
    for groups in (3, 11):
        k1 = rng.choice(np.array([-7, 3, 1 << 20, 5, 6, 9][:max(2, groups // 2)], dtype=np.int32), size=n)      # dictionary-coded
        k2 = rng.integers(0, max(2, groups // 3), size=n).astype(np.int64)                                       # truncated
        qty = rng.integers(1, 51, size=n).astype(np.float64)
        cnt = rng.integers(0, 200, size=n).astype(np.int32)
        price = np.round(rng.uniform(900, 105000, size=n), 2)
        steps = rng.integers(0, 150, size=n) * 0.5 - 7.0                                                         # a 150-entry dictionary
        cols = [k1, k2, qty, cnt, price, steps]
        comp = [oracle.CompressedColumn(c) for c in cols[:4]] + [None, oracle.CompressedColumn(steps)]
        widths = [c.code_width if c is not None and c.kind != 0 else 0 for c in comp]
        assert widths[0] == 1 and comp[0].dictionary is not None and widths[1] == 1 and comp[1].dictionary is None and widths[4] == 0
        assert widths[5] == 1 and comp[5].dictionary is not None and comp[5].dictionary.size == 150 and comp[2].dictionary.size == 50
        layout = [(T.INT, None), (T.LONG, None), (T.DOUBLE, None), (T.INT, None), (T.DOUBLE, None), (T.DOUBLE, None)]
        kw = dict(keys=[0, 1], instrs=[(T.EX_MUL, 0, T.col(2), T.col(4))],
                  aggs=[(T.AGG_SUM, T.col(2)), (T.AGG_SUM, T.temp(0)), (T.AGG_MIN, T.col(4)), (T.AGG_COUNT_STAR, None),
                        (T.AGG_SUM, T.col(5))],      # (+ the four hidden accumulators of the 12-byte key: the state's eight)
                  pred=[(3, T.GE, 20), (0, T.NE, 5)], est_groups=4)
        cfg = T.make_agg_config(T.AGG_GENERIC, layout, code_widths=widths, **kw)
        code_cols = [cols[i] if widths[i] == 0 else comp[i].codes for i in range(6)]
        dicts = [None if widths[i] == 0 else comp[i].dictionary for i in range(6)]
        st = capi.AggState(cfg)
        o = oracle.AggState(cfg)
        filt = oracle.bitmap_from_bools(rng.random(n) < 0.6)
        for lo, hi, use_filter in ((0, 100_000, False), (100_000, n, True)):
            f = oracle.bitmap_from_bools(oracle.bools_from_bitmap(filt, n)[lo:hi]) if use_filter else None
            st.update_coded([to_dev(np.ascontiguousarray(c[lo:hi]), dev) for c in code_cols],
                            [None if d is None else to_dev(d, dev) for d in dicts], hi - lo,
                            filter_bitmap=None if f is None else bitmap_dev(f, dev), sized=kernel != "plan_shape_unsized_dictionaries")
            o.update_coded([np.ascontiguousarray(c[lo:hi]) for c in code_cols], dicts, hi - lo, filter_bitmap=f)
        assert_same_groups(finalize_np(st, dev), o.finalize())
        if kernel != "interpreter":
            assert capi.lib.qsx_debug_agg_jit_state(st._h, 1) == 1       # (the filtered variant ran as a compiled shape)


def test_dense_group_by_over_a_truncated_key(capi, oracle, dev):
    """COLLISION_FREE state whose key attribute is truncation-compressed (2-byte codes of an INT) and whose argument is
    dictionary-coded: keys are decoded while staging like every other column."""
    rng = np.random.default_rng(5)
    n = 300_000
    key = rng.integers(0, 40_000, size=n).astype(np.int32)
    val = rng.choice(np.array([0.25, 1.5, -3.0, 1e6, 7.0]), size=n)
    ck, cv = oracle.CompressedColumn(key), oracle.CompressedColumn(val)
    assert ck.code_width == 2 and ck.dictionary is None and cv.code_width == 1 and cv.dictionary is not None
    cfg = T.make_agg_config(T.AGG_COLLISION_FREE, [(T.INT, None), (T.DOUBLE, None)], keys=[0],
                            aggs=[(T.AGG_SUM, T.col(1)), (T.AGG_COUNT_STAR, None), (T.AGG_MIN, T.col(1))], num_entries=40_000,
                            code_widths=[2, 1])
    st = capi.AggState(cfg)
    st.update_coded([codes_dev(ck, dev), codes_dev(cv, dev)], [None, to_dev(cv.dictionary, dev)], n)
    o = oracle.AggState(cfg)
    o.update_coded([ck.codes, cv.codes], [None, cv.dictionary], n)
    from test_gpu_agg import assert_same_groups, finalize_np
    assert_same_groups(finalize_np(st, dev), o.finalize())


def test_aggregation_through_a_pair_list(capi, oracle, dev):
    """The coded entry point doubles as a selection vector: a column handed over as 4-byte "codes" = row numbers (the
    probe tids of a join) with the column itself as dictionary is read through the pair list — the join output is never
    materialised.  Same groups as gather-then-aggregate (bit-exact: same values in the same order per group)."""
    rng = np.random.default_rng(33)
    n_rows, n_pairs = 500_000, 120_000
    orderkey = np.sort(rng.integers(0, 60_000, size=n_rows)).astype(np.int32)
    price = np.round(rng.uniform(900, 105000, size=n_rows), 2)
    disc = rng.integers(0, 11, size=n_rows) / 100.0
    tids = np.sort(rng.choice(n_rows, size=n_pairs, replace=False)).astype(np.int32)      # ascending, like the two-pass probe emits
    layout = [(T.INT, None), (T.DOUBLE, None), (T.DOUBLE, None)]
    kw = dict(keys=[0], instrs=[(T.EX_SUB, 0, T.const(0), T.col(2)), (T.EX_MUL, 1, T.col(1), T.temp(0))], consts=[1.0],
              aggs=[(T.AGG_SUM, T.temp(1)), (T.AGG_COUNT_STAR, None)], num_entries=60_000)
    through = capi.AggState(T.make_agg_config(T.AGG_COLLISION_FREE, layout, code_widths=[4, 4, 4], **kw))
    d_tids = to_dev(tids, dev)
    through.update_coded([d_tids, d_tids, d_tids], [to_dev(orderkey, dev), to_dev(price, dev), to_dev(disc, dev)], n_pairs)
    o = oracle.AggState(T.make_agg_config(T.AGG_COLLISION_FREE, layout, **kw))
    o.update([orderkey[tids], price[tids], disc[tids]])
    from test_gpu_agg import assert_same_groups, finalize_np
    assert_same_groups(finalize_np(through, dev), o.finalize())


@pytest.mark.parametrize("jit", [False, True])
def test_q1_over_a_run_of_compressed_blocks(capi, oracle, dev, jit, monkeypatch):
    """qsx_agg_update_coded_blocks: one launch over a run of compressed blocks, every block compressed on its own (its own
    dictionaries: a block that holds fewer distinct values has a different code assignment), ragged sizes, filters with gaps —
    equal to the oracle aggregating the decoded blocks one by one."""
    monkeypatch.setenv("QSX_AGG_JIT_MIN_ROWS", "0" if jit else str(1 << 60))
    rng = np.random.default_rng(78)
    rows = [70_001, 0, 20_513, 31_024, 150_000, 9_005]
    layout = [(T.CHAR, 1), (T.CHAR, 1), (T.DOUBLE, None), (T.DOUBLE, None), (T.DOUBLE, None), (T.DOUBLE, None), (T.INT, None)]
    kw = dict(keys=[0, 1],
              instrs=[(T.EX_SUB, 0, T.const(0), T.col(4)), (T.EX_MUL, 1, T.col(3), T.temp(0)),
                      (T.EX_ADD, 2, T.const(0), T.col(5)), (T.EX_MUL, 3, T.temp(1), T.temp(2))],
              consts=[1.0],
              aggs=[(T.AGG_SUM, T.col(2)), (T.AGG_SUM, T.col(3)), (T.AGG_SUM, T.temp(1)), (T.AGG_SUM, T.temp(3)),
                    (T.AGG_AVG, T.col(4)), (T.AGG_COUNT_STAR, None), (T.AGG_SUM, T.col(6)), (T.AGG_MAX, T.col(4))],
              pred=[(6, T.LT, 7)], est_groups=6)
    widths = [0, 0, 1, 0, 1, 1, 1]
    cfg = T.make_agg_config(T.AGG_COMPACT_KEY, layout, code_widths=widths, **kw)
    plain_cfg = T.make_agg_config(T.AGG_COMPACT_KEY, layout, **kw)
    st = capi.AggState(cfg)
    o = oracle.AggState(plain_cfg)
    blocks, dicts, filters = [], [], []
    for b, n in enumerate(rows):
        cols, comp = _q1_coded_inputs(oracle, rng, n) if n else ([np.zeros(0, dtype=t) for t in (np.uint8, np.uint8, np.float64, np.float64, np.float64, np.float64, np.int32)], None)
        if n:
            if b % 2 == 0:
                cols[2] = np.where(cols[2] > 25, 25.0, cols[2])      # fewer distinct quantities in this block: another dictionary
                comp[2] = oracle.CompressedColumn(cols[2])
            assert [0 if c is None or c.kind == 0 else c.code_width for c in comp] == widths
            blocks.append([to_dev(cols[i] if widths[i] == 0 else comp[i].codes, dev) for i in range(7)])
            dicts.append([None if widths[i] == 0 or comp[i].dictionary is None else to_dev(comp[i].dictionary, dev) for i in range(7)])
        else:
            blocks.append([to_dev(c, dev) for c in cols])
            dicts.append([None] * 7)
        f = oracle.bitmap_from_bools(rng.random(n) < 0.8) if (b % 3 == 0 and n) else None
        filters.append(None if f is None else bitmap_dev(f, dev))
        if n:
            o.update(cols, filter_bitmap=f)
    st.update_coded_blocks(blocks, dicts, filters)
    from test_gpu_agg import assert_same_groups, finalize_np
    assert_same_groups(finalize_np(st, dev), o.finalize())
    with pytest.raises(capi.QsxError):                               # a coded state needs the coded entry point
        st.update_blocks(blocks)


def test_sort_column_predicates_over_runs_of_blocks(capi, oracle, dev):
    """qsx_select_cmp_sorted_blocks / qsx_select_codes_sorted_blocks: every block sorted on its own, one launch for the run,
    per-block comparisons on the code stripes of a compressed sort column — equal to the scan of every block."""
    rng = np.random.default_rng(31)
    rows = [5000, 0, 1, 63, 64, 65, 4096, 4097, 70_001, 0, 33]
    # uncompressed sort column, every type
    for np_t, qt in ((np.int32, None), (np.int64, None), (np.float32, None), (np.float64, None)):
        blocks = [np.sort(rng.integers(-50, 50, size=n)).astype(np_t) for n in rows]
        filters = [oracle.bitmap_from_bools(rng.random(n) < 0.6) if (i % 3 != 1 and n) else None for i, n in enumerate(rows)]
        dblocks = [to_dev(b, dev) for b in blocks]
        for op in (T.EQ, T.NE, T.LT, T.LE, T.GT, T.GE):
            for lit in (3, 60):
                for use_filters in (False, True):
                    outs, counts = capi.select_cmp_sorted_blocks(dblocks, op, lit, filters=[None if f is None else bitmap_dev(f, dev) for f in filters] if use_filters else None)
                    for b, n in enumerate(rows):
                        want = oracle.select_cmp(blocks[b], op, np_t(lit), filter_bitmap=filters[b] if use_filters else None)
                        if n:
                            assert np.array_equal(bitmap_np(outs[b])[:want.size], want), (np_t, op, lit, b, n)
                        assert int(counts[b].item()) == oracle.bitmap_count(want, n)
    # DATE sort column
    from test_gpu_select import make_dates
    dates = [make_dates(rng, n) for n in (3000, 0, 70_001, 5)]
    dates = [np.ascontiguousarray(d[oracle.sort_permutation([d], types=[T.DATE])]) if d.size else d for d in dates]
    lit = T.date_raw(1995, 3, 15)
    for op in (T.EQ, T.LT, T.GE, T.NE):
        outs, counts = capi.select_cmp_sorted_blocks([to_dev(d, dev) for d in dates], op, lit, qtype=T.DATE)
        for i, d in enumerate(dates):
            want = oracle.select_cmp(d, op, lit, qt=T.DATE)
            if d.size:
                assert np.array_equal(bitmap_np(outs[i])[:want.size], want)
            assert int(counts[i].item()) == oracle.bitmap_count(want, d.size)
    # compressed sort column: per-block comparisons on the code stripes
    for width, np_t, hi in ((1, np.uint8, 256), (2, np.uint16, 65536), (4, np.uint32, 2**32)):
        code_blocks = [np.sort(rng.integers(0, min(hi, max(4, n // 3 + 4)), size=n, dtype=np.uint64)).astype(np_t) for n in rows]
        dcodes = [torch.from_numpy(c.view({1: np.uint8, 2: np.int16, 4: np.int32}[width]).copy()).to(dev) for c in code_blocks]
        ops = [(T.CODE_EQ, T.CODE_NE, T.CODE_LT, T.CODE_GE, T.CODE_RANGE)[i % 5] for i in range(len(rows))]
        firsts = [int(c[c.size // 2]) if c.size else 0 for c in code_blocks]
        seconds = [min(hi - 1, f + 5) for f in firsts]
        filters = [oracle.bitmap_from_bools(rng.random(n) < 0.6) if (i % 2 == 0 and n) else None for i, n in enumerate(rows)]
        outs, counts = capi.select_codes_sorted_blocks(dcodes, ops, firsts, seconds, filters=[None if f is None else bitmap_dev(f, dev) for f in filters])
        for b, n in enumerate(rows):
            want = oracle.select_codes(code_blocks[b], ops[b], firsts[b], seconds[b], filters[b])
            if n:
                assert np.array_equal(bitmap_np(outs[b])[:want.size], want), (width, b, n, ops[b])
            assert int(counts[b].item()) == oracle.bitmap_count(want, n)


def test_code_stripe_scans_over_runs_of_blocks(capi, oracle, dev):
    """qsx_select_codes_blocks: one launch over the code stripes of a run of blocks with the comparison given per block
    (every block has its own dictionary) — equal to qsx_select_codes' scan block by block, also for unaligned stripes."""
    rng = np.random.default_rng(32)
    rows = [5000, 0, 1, 63, 64, 65, 4095, 4096, 4097, 70_001, 0, 33]
    all_ops = (T.CODE_EQ, T.CODE_NE, T.CODE_LT, T.CODE_GE, T.CODE_RANGE)
    for width, np_t, hi in ((1, np.uint8, 256), (2, np.uint16, 65536), (4, np.uint32, 2**32)):
        torch_view = {1: np.uint8, 2: np.int16, 4: np.int32}[width]
        for shift in (0, 1):                                            # 1: every stripe starts inside a 16-byte chunk
            blocks = [rng.integers(0, min(hi, 40), size=n + shift, dtype=np.uint64).astype(np_t) for n in rows]
            dcodes = [torch.from_numpy(c.view(torch_view).copy()).to(dev)[shift:] for c in blocks]
            blocks = [c[shift:] for c in blocks]
            for rot in range(2):
                ops = [all_ops[(i + rot) % 5] for i in range(len(rows))]
                firsts = [int(rng.integers(0, 40)) for _ in rows]
                seconds = [f + int(rng.integers(0, 9)) for f in firsts]
                filters = [oracle.bitmap_from_bools(rng.random(n) < 0.6) if (i % 2 == rot and n) else None for i, n in enumerate(rows)]
                outs, counts = capi.select_codes_blocks(dcodes, ops, firsts, seconds,
                                                        filters=[None if f is None else bitmap_dev(f, dev) for f in filters])
                for b, n in enumerate(rows):
                    want = oracle.select_codes(blocks[b], ops[b], firsts[b], seconds[b], filters[b])
                    if n:
                        assert np.array_equal(bitmap_np(outs[b])[:want.size], want), (width, shift, b, n, ops[b])
                    assert int(counts[b].item()) == oracle.bitmap_count(want, n)


def test_coded_plan_shapes_built_by_hiprtc_match_too():
    """The in-process compiler (QSX_JIT_COMPILER=hiprtc: what builds the shapes on a host without hipcc) produces code of its
    own; the coded aggregation tests of this file — decoded columns in keys, predicates, integer and floating sums — once
    passed with the driver's build and returned garbage with hipRTC's (a register struct in scratch).  Run them again in a
    process that only has hipRTC."""
    import subprocess
    import sys
    env = dict(os.environ, QSX_JIT_COMPILER="hiprtc")
    nodes = ["test_q1_over_compressed_attributes_matches_oracle", "test_group_by_compressed_keys_matches_oracle",
             "test_q1_over_a_run_of_compressed_blocks"]
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-k", " or ".join(nodes)],
                       env=env, capture_output=True, text=True, timeout=900, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def _factored_launches(capi):
    import ctypes
    fn = capi.lib.qsx_debug_agg_factored_launches
    fn.restype = ctypes.c_longlong
    return fn()


def _coded(oracle, values, dtype=None):
    """(codes, dictionary) of a column as CompressedBlockBuilder stores it when it picks a dictionary."""
    col = oracle.CompressedColumn(values if dtype is None else values.astype(dtype))
    assert col.dictionary is not None, "the test wants a dictionary-coded column"
    return col


_FACTORED_PLANS = {
    # TPC-H Q1 (benchmarks/tpch/queries/01.sql) over lineitem's code stripes: cells (disc, tax), a histogram for quantity
    "q1": dict(instrs=[(T.EX_SUB, 0, T.const(0), T.col(4)), (T.EX_MUL, 1, T.col(3), T.temp(0)), (T.EX_ADD, 2, T.const(0), T.col(5)),
                       (T.EX_MUL, 3, T.temp(1), T.temp(2))], consts=[1.0],
               aggs=[(T.AGG_SUM, T.col(2)), (T.AGG_SUM, T.col(3)), (T.AGG_SUM, T.temp(1)), (T.AGG_SUM, T.temp(3)), (T.AGG_AVG, T.col(2)),
                     (T.AGG_AVG, T.col(3)), (T.AGG_AVG, T.col(4)), (T.AGG_COUNT_STAR, None)], factored=True),
    # affine forms: price + disc, disc * tax (two dictionary columns, no plain one), price / (1 + tax), a constant-free product
    "affine": dict(instrs=[(T.EX_ADD, 0, T.col(3), T.col(4)), (T.EX_MUL, 1, T.col(4), T.col(5)), (T.EX_ADD, 2, T.const(0), T.col(5)),
                           (T.EX_DIV, 3, T.col(3), T.temp(2)), (T.EX_SUB, 4, T.temp(3), T.col(6))], consts=[1.0],
                   aggs=[(T.AGG_SUM, T.temp(0)), (T.AGG_SUM, T.temp(1)), (T.AGG_SUM, T.temp(3)), (T.AGG_AVG, T.temp(4)), (T.AGG_COUNT_STAR, None)],
                   factored=True),
    # integer sums: over a dictionary-coded INT column (histogram, exact), over a plain INT column (an i64 plane), next to a double sum
    "integers": dict(instrs=[(T.EX_MUL, 0, T.col(3), T.col(4))], consts=[],
                     aggs=[(T.AGG_SUM, T.col(7)), (T.AGG_SUM, T.col(8)), (T.AGG_SUM, T.temp(0)), (T.AGG_AVG, T.col(7)), (T.AGG_COUNT_STAR, None)], factored=True),
    # two plain columns (price and a FLOAT) times dictionary factors
    "two_carriers": dict(instrs=[(T.EX_MUL, 0, T.col(3), T.col(4)), (T.EX_MUL, 1, T.col(6), T.col(5)), (T.EX_ADD, 2, T.temp(0), T.temp(1))], consts=[],
                         aggs=[(T.AGG_SUM, T.temp(2)), (T.AGG_SUM, T.col(6)), (T.AGG_COUNT_STAR, None)], factored=True),
    # not affine in the plain column: the decoding kernels answer
    "square": dict(instrs=[(T.EX_MUL, 0, T.col(3), T.col(3)), (T.EX_MUL, 1, T.temp(0), T.col(4))], consts=[],
                   aggs=[(T.AGG_SUM, T.temp(1)), (T.AGG_COUNT_STAR, None)], factored=False),
    # MIN / MAX do not factor
    "minmax": dict(instrs=[], consts=[], aggs=[(T.AGG_SUM, T.col(4)), (T.AGG_MAX, T.col(3)), (T.AGG_COUNT_STAR, None)], factored=False),
}


@pytest.mark.parametrize("plan", sorted(_FACTORED_PLANS))
@pytest.mark.parametrize("groups,est", [(4, 6), (40, 2)])
def test_aggregates_factored_through_dictionary_codes_match_the_oracle(capi, oracle, dev, plan, groups, est, monkeypatch):
    """csrc/agg_factored.hpp: a state over dictionary-coded attributes whose aggregate arguments are affine in the plain
    columns adds every row to the CELL (group, codes of the dictionary columns) — carrier sums and a count — and to a
    histogram per stand-alone dictionary column; the state's accumulators are dot products of the cells with coefficients made
    from the call's dictionaries.  Against the oracle (which decodes every value, like the reference's accessor): COUNT and
    integer sums exact, SUM over an integer-valued dictionary column exact, the rest to 1e-6 (north_star) — under a filter,
    over a tail that is not a multiple of the tile, with more groups than the workgroup's table holds (est = 2: the rows of
    the other groups take the per-row path), and for plans that do NOT factor (answered by the decoding kernels)."""
    monkeypatch.setenv("QSX_AGG_FACTORED_MIN_ROWS", "0")
    monkeypatch.setenv("QSX_AGG_FACTORED_GENERIC", "1")            # keys of two widths: none of the direct-load signatures, the staged kernel answers
    monkeypatch.setenv("QSX_AGG_JIT_MIN_ROWS", str(1 << 60))        # (the fall-back, when it is taken, is the interpreter: no compile in this test)
    rng = np.random.default_rng(500 + groups)
    n = 300_007
    k1 = rng.integers(0, groups, size=n).astype(np.int32) * 7 - 3
    k2 = rng.choice(np.frombuffer(b"FO", dtype=np.uint8), size=n)
    qty = rng.integers(1, 51, size=n).astype(np.float64)
    price = np.round(rng.uniform(900, 105000, size=n), 2)
    disc = rng.integers(0, 11, size=n) / 100.0
    tax = rng.integers(0, 9, size=n) / 100.0
    weight = rng.uniform(-2, 2, size=n).astype(np.float32)
    size_code = (rng.integers(0, 40, size=n) * 1000 - 7000).astype(np.int32)      # a dictionary-coded INT (negative values: no truncation)
    plain_int = rng.integers(-10**6, 10**6, size=n).astype(np.int32)
    cols = [k1, k2, qty, price, disc, tax, weight, size_code, plain_int]
    comp = {2: _coded(oracle, qty), 4: _coded(oracle, disc), 5: _coded(oracle, tax), 7: _coded(oracle, size_code)}
    widths = [comp[i].code_width if i in comp else 0 for i in range(len(cols))]
    assert widths[2] == widths[4] == widths[5] == widths[7] == 1
    layout = [(T.INT, None), (T.CHAR, 1), (T.DOUBLE, None), (T.DOUBLE, None), (T.DOUBLE, None), (T.DOUBLE, None), (T.FLOAT, None), (T.INT, None), (T.INT, None)]
    spec = _FACTORED_PLANS[plan]
    cfg = T.make_agg_config(T.AGG_COMPACT_KEY, layout, keys=[0, 1], instrs=spec["instrs"], consts=spec["consts"], aggs=spec["aggs"],
                            est_groups=est, code_widths=widths)
    code_cols = [comp[i].codes if i in comp else cols[i] for i in range(len(cols))]
    dicts = [comp[i].dictionary if i in comp else None for i in range(len(cols))]
    filt = oracle.bitmap_from_bools(rng.random(n) < 0.75)
    before = _factored_launches(capi)
    st, o = capi.AggState(cfg), oracle.AggState(cfg)
    cut = 131_072 + 5
    for lo, hi, use_filter in ((0, cut, False), (cut, n, True)):
        f = oracle.bitmap_from_bools(oracle.bools_from_bitmap(filt, n)[lo:hi]) if use_filter else None
        st.update_coded([to_dev(np.ascontiguousarray(c[lo:hi]), dev) for c in code_cols], [None if d is None else to_dev(d, dev) for d in dicts],
                        hi - lo, filter_bitmap=None if f is None else bitmap_dev(f, dev))
        o.update_coded([np.ascontiguousarray(c[lo:hi]) for c in code_cols], dicts, hi - lo, filter_bitmap=f)
    assert (_factored_launches(capi) - before == 2) == spec["factored"], "the plan took the other path"
    from test_gpu_agg import assert_same_groups, finalize_np
    assert_same_groups(finalize_np(st, dev), o.finalize())
    # a call without the dictionaries' sizes cannot factor (the cells' extent is unknown): same groups from the decoding kernels
    st2 = capi.AggState(cfg)
    st2.update_coded([to_dev(c, dev) for c in code_cols], [None if d is None else to_dev(d, dev) for d in dicts], n, sized=False)
    o2 = oracle.AggState(cfg)
    o2.update_coded(code_cols, dicts, n)
    assert_same_groups(finalize_np(st2, dev), o2.finalize())


@pytest.mark.parametrize("key_kind", ["char", "int"])
@pytest.mark.parametrize("shape", ["q1", "one_cell_no_histogram", "two_cells_no_carrier"])
@pytest.mark.parametrize("groups,est", [(3, 6), (50, 2)])
def test_factored_aggregation_by_direct_loads_matches_the_oracle(capi, oracle, dev, key_kind, shape, groups, est, monkeypatch):
    """agg_factored_direct_kernel (csrc/agg_factored_kernels.hpp): the signatures answered without LDS staging — one or two
    keys of one width (CHAR(1) or INT), one or two cell columns, at most one histogram column, at most one DOUBLE carrier;
    8 consecutive rows per thread by 8- / 16-byte loads, the stripe's tail row by row.  Q1 over lineitem's codes is the
    first of them (benchmarks/tpch/queries/01.sql over create.sql:69-121).  Against the oracle, with and without a filter, a
    row count that is no multiple of anything, a slice that starts at an odd tile, and more groups than the table holds."""
    monkeypatch.setenv("QSX_AGG_FACTORED_MIN_ROWS", "0")
    monkeypatch.setenv("QSX_AGG_JIT_MIN_ROWS", str(1 << 60))
    monkeypatch.delenv("QSX_AGG_FACTORED_GENERIC", raising=False)
    rng = np.random.default_rng(900 + groups)
    n = 2048 * 37 + 1234                                              # 37 full tiles of the direct kernel and a tail
    if key_kind == "char":
        letters = np.frombuffer(b"ABCDEFGHIJKLMNOPQRSTUVWXYZ", dtype=np.uint8)
        k1, k2 = rng.choice(letters[:max(1, groups // 2)], size=n), rng.choice(letters[:2], size=n)
        key_layout = [(T.CHAR, 1), (T.CHAR, 1)]
    else:
        k1 = (rng.integers(0, max(1, groups // 2), size=n) * 1009 - 5).astype(np.int32)
        k2 = rng.integers(0, 2, size=n).astype(np.int32)
        key_layout = [(T.INT, None), (T.INT, None)]
    qty = rng.integers(1, 51, size=n).astype(np.float64)
    price = np.round(rng.uniform(900, 105000, size=n), 2)
    disc = rng.integers(0, 11, size=n) / 100.0
    tax = rng.integers(0, 9, size=n) / 100.0
    cols = [k1, k2, qty, price, disc, tax]
    comp = {2: _coded(oracle, qty), 4: _coded(oracle, disc), 5: _coded(oracle, tax)}
    widths = [comp[i].code_width if i in comp else 0 for i in range(len(cols))]
    layout = key_layout + [(T.DOUBLE, None)] * 4
    if shape == "q1":
        spec = _FACTORED_PLANS["q1"]
        keys = [0, 1]
    elif shape == "one_cell_no_histogram":     # SUM(price * (1 - disc)), AVG(disc): one key, one cell column, one carrier
        spec = dict(instrs=[(T.EX_SUB, 0, T.const(0), T.col(4)), (T.EX_MUL, 1, T.col(3), T.temp(0))], consts=[1.0],
                    aggs=[(T.AGG_SUM, T.temp(1)), (T.AGG_AVG, T.col(4)), (T.AGG_COUNT_STAR, None)])
        keys = [0]
    else:                                      # SUM(disc * tax), SUM(qty): two cell columns, a histogram, no plain column at all
        spec = dict(instrs=[(T.EX_MUL, 0, T.col(4), T.col(5))], consts=[], aggs=[(T.AGG_SUM, T.temp(0)), (T.AGG_SUM, T.col(2)), (T.AGG_COUNT_STAR, None)])
        keys = [0, 1]
    cfg = T.make_agg_config(T.AGG_COMPACT_KEY, layout, keys=keys, instrs=spec["instrs"], consts=spec["consts"], aggs=spec["aggs"],
                            est_groups=est, code_widths=widths)
    code_cols = [comp[i].codes if i in comp else cols[i] for i in range(len(cols))]
    dicts = [comp[i].dictionary if i in comp else None for i in range(len(cols))]
    filt = oracle.bitmap_from_bools(rng.random(n) < 0.7)
    dev_cols = [to_dev(c, dev) for c in code_cols]
    dev_dicts = [None if d is None else to_dev(d, dev) for d in dicts]
    before = _factored_launches(capi)
    st, o = capi.AggState(cfg), oracle.AggState(cfg)
    st.update_coded(dev_cols, dev_dicts, n)
    o.update_coded(code_cols, dicts, n)
    st.update_coded(dev_cols, dev_dicts, n, filter_bitmap=bitmap_dev(filt, dev))
    o.update_coded(code_cols, dicts, n, filter_bitmap=filt)
    lo = 2048 * 3                                                     # a slice (16-byte aligned for every column width)
    st.update_coded([c[lo:] for c in dev_cols], dev_dicts, n - lo)
    o.update_coded([np.ascontiguousarray(c[lo:]) for c in code_cols], dicts, n - lo)
    assert _factored_launches(capi) - before == 3, "the signature did not take the direct-load kernel"
    from test_gpu_agg import assert_same_groups, finalize_np
    assert_same_groups(finalize_np(st, dev), o.finalize())


@pytest.mark.parametrize("key_kind", ["char", "int"])
@pytest.mark.parametrize("shape", ["q1", "one_cell_no_histogram"])
@pytest.mark.parametrize("groups,est", [(3, 6), (50, 2)])
def test_factored_aggregation_over_a_run_of_blocks_with_their_own_dictionaries(capi, oracle, dev, key_kind, shape, groups, est, monkeypatch):
    """qsx_agg_update_coded_blocks_sized: the reference compresses block by block (storage/CompressedBlockBuilder.cpp:300-368), so
    one attribute's codes mean other values in every block — here every block draws its discounts, taxes and quantities from a
    different subset, its dictionaries differ in size and content.  The factored kernel (agg_factored_direct_kernel, run form)
    walks a contiguous range of the run's tiles per workgroup and settles its per-code cells with the block's coefficient
    tables at every block boundary.  Blocks of fewer rows than a tile, of exactly whole tiles, with tails that are no multiple
    of 8; filters on some blocks only; more groups than the workgroup's table holds.  Against the oracle block by block."""
    monkeypatch.setenv("QSX_AGG_FACTORED_MIN_ROWS", "0")
    monkeypatch.setenv("QSX_AGG_JIT_MIN_ROWS", str(1 << 60))
    monkeypatch.delenv("QSX_AGG_FACTORED_GENERIC", raising=False)
    rng = np.random.default_rng(1300 + groups)
    sizes = [2048 * 3 + 5, 300, 2048 * 9, 70_001, 517, 2048, 33_333, 2048 * 2 + 2047]
    layout = ([(T.CHAR, 1), (T.CHAR, 1)] if key_kind == "char" else [(T.INT, None), (T.INT, None)]) + [(T.DOUBLE, None)] * 4
    if shape == "q1":
        spec, keys = _FACTORED_PLANS["q1"], [0, 1]
    else:
        spec = dict(instrs=[(T.EX_SUB, 0, T.const(0), T.col(4)), (T.EX_MUL, 1, T.col(3), T.temp(0))], consts=[1.0],
                    aggs=[(T.AGG_SUM, T.temp(1)), (T.AGG_AVG, T.col(4)), (T.AGG_COUNT_STAR, None)])
        keys = [0]
    blocks, block_dicts, filters, host = [], [], [], []
    widths = None
    for b, n in enumerate(sizes):
        if key_kind == "char":
            letters = np.frombuffer(b"ABCDEFGHIJKLMNOPQRSTUVWXYZ", dtype=np.uint8)
            k1, k2 = rng.choice(letters[:max(1, groups // 2)], size=n), rng.choice(letters[:2], size=n)
        else:
            k1 = (rng.integers(0, max(1, groups // 2), size=n) * 1009 - 5).astype(np.int32)
            k2 = rng.integers(0, 2, size=n).astype(np.int32)
        # every block its own subsets: dictionaries of 2 .. 50 / 11 / 9 entries whose code i is another value from block to block
        qty_values = rng.choice(np.arange(1, 51), size=int(rng.integers(2, 51)), replace=False).astype(np.float64)
        disc_values = rng.choice(np.arange(0, 11), size=int(rng.integers(2, 12)), replace=False) / 100.0
        tax_values = rng.choice(np.arange(0, 9), size=int(rng.integers(2, 10)), replace=False) / 100.0
        qty, disc, tax = rng.choice(qty_values, size=n), rng.choice(disc_values, size=n), rng.choice(tax_values, size=n)
        price = np.round(rng.uniform(900, 105000, size=n), 2)
        cols = [k1, k2, qty, price, disc, tax]
        comp = {2: _coded(oracle, qty), 4: _coded(oracle, disc), 5: _coded(oracle, tax)}
        w = [comp[i].code_width if i in comp else 0 for i in range(len(cols))]
        widths = widths or w
        assert w == widths
        code_cols = [comp[i].codes if i in comp else cols[i] for i in range(len(cols))]
        dicts = [comp[i].dictionary if i in comp else None for i in range(len(cols))]
        filt = oracle.bitmap_from_bools(rng.random(n) < 0.6) if b % 3 == 1 else None
        host.append((code_cols, dicts, n, filt))
        blocks.append([to_dev(c, dev) for c in code_cols])
        block_dicts.append([None if d is None else to_dev(d, dev) for d in dicts])
        filters.append(None if filt is None else bitmap_dev(filt, dev))
    cfg = T.make_agg_config(T.AGG_COMPACT_KEY, layout, keys=keys, instrs=spec["instrs"], consts=spec["consts"], aggs=spec["aggs"],
                            est_groups=est, code_widths=widths)
    from test_gpu_agg import assert_same_groups, finalize_np
    before = _factored_launches(capi)
    st, o = capi.AggState(cfg), oracle.AggState(cfg)
    st.update_coded_blocks(blocks, block_dicts, filters)               # some blocks filtered
    # no filter at all, and a block without tuples in the middle of the run (it has stripes and dictionaries, and no rows)
    empty = [c[:0] for c in blocks[0]]
    st.update_coded_blocks([blocks[2], blocks[3], empty, blocks[4]], [block_dicts[2], block_dicts[3], block_dicts[0], block_dicts[4]])
    for code_cols, dicts, n, filt in host:
        o.update_coded(code_cols, dicts, n, filter_bitmap=filt)
    for code_cols, dicts, n, filt in host[2:5]:
        o.update_coded(code_cols, dicts, n)
    assert _factored_launches(capi) - before == 2, "the run did not take the factored kernel"
    assert_same_groups(finalize_np(st, dev), o.finalize())
    # without the sizes the decoding kernels answer: same groups
    st2 = capi.AggState(cfg)
    st2.update_coded_blocks(blocks, block_dicts, filters, sized=False)
    assert _factored_launches(capi) - before == 2
    o2 = oracle.AggState(cfg)
    for code_cols, dicts, n, filt in host:
        o2.update_coded(code_cols, dicts, n, filter_bitmap=filt)
    assert_same_groups(finalize_np(st2, dev), o2.finalize())


class _StripeAt:
    """A stripe at an arbitrary byte address inside a device buffer (what a reference block image holds: offsets that are
    multiples of the block's tuple capacity, aligned to nothing)."""

    def __init__(self, values, shift, dev):
        raw = np.ascontiguousarray(values).view(np.uint8).reshape(-1)
        self.buffer = torch.zeros(raw.size + 64, dtype=torch.uint8, device=dev)
        self.buffer[shift:shift + raw.size] = torch.from_numpy(raw.copy()).to(dev)
        self.shift, self.rows = shift, len(values)

    def data_ptr(self):
        return self.buffer.data_ptr() + self.shift

    def numel(self):
        return self.rows


@pytest.mark.parametrize("shifts", [(1, 3, 5, 7, 9, 11), (0, 0, 13, 4, 0, 6), (16, 8, 2, 1, 3, 15)])
def test_factored_aggregation_reads_stripes_at_any_byte_address(capi, oracle, dev, shifts, monkeypatch):
    """A reference CompressedColumnStore block keeps attribute stripes back to back at multiples of its tuple capacity
    (storage/CompressedColumnStoreTupleStorageSubBlock.cpp:71-160): a DOUBLE stripe can start at an odd address.  The direct-load
    factored kernel reads 8 rows per thread with 8- and 16-byte loads at whatever address that is — one stripe and a run of
    blocks, every stripe shifted by another number of bytes.  Against the oracle."""
    monkeypatch.setenv("QSX_AGG_FACTORED_MIN_ROWS", "0")
    monkeypatch.setenv("QSX_AGG_JIT_MIN_ROWS", str(1 << 60))
    rng = np.random.default_rng(77)
    spec = _FACTORED_PLANS["q1"]
    layout = [(T.CHAR, 1), (T.CHAR, 1)] + [(T.DOUBLE, None)] * 4
    from test_gpu_agg import assert_same_groups, finalize_np
    cfg = None
    st = o = None
    before = _factored_launches(capi)
    blocks, block_dicts, keep = [], [], []
    for n in (2048 * 5 + 77, 40_001, 2048 * 2):
        letters = np.frombuffer(b"ANR", dtype=np.uint8)
        cols = [rng.choice(letters, size=n), rng.choice(np.frombuffer(b"FO", dtype=np.uint8), size=n), rng.integers(1, 51, size=n).astype(np.float64),
                np.round(rng.uniform(900, 105000, size=n), 2), rng.integers(0, 11, size=n) / 100.0, rng.integers(0, 9, size=n) / 100.0]
        comp = {2: _coded(oracle, cols[2]), 4: _coded(oracle, cols[4]), 5: _coded(oracle, cols[5])}
        widths = [comp[i].code_width if i in comp else 0 for i in range(6)]
        if cfg is None:
            cfg = T.make_agg_config(T.AGG_COMPACT_KEY, layout, keys=[0, 1], instrs=spec["instrs"], consts=spec["consts"], aggs=spec["aggs"],
                                    est_groups=6, code_widths=widths)
            st, o = capi.AggState(cfg), oracle.AggState(cfg)
        code_cols = [comp[i].codes if i in comp else cols[i] for i in range(6)]
        dicts = [comp[i].dictionary if i in comp else None for i in range(6)]
        shifted = [_StripeAt(c, sh, dev) for c, sh in zip(code_cols, shifts)]
        dev_dicts = [None if d is None else to_dev(d, dev) for d in dicts]
        keep.append(shifted)
        st.update_coded(shifted, dev_dicts, n)                       # one stripe
        o.update_coded(code_cols, dicts, n)
        blocks.append(shifted)
        block_dicts.append(dev_dicts)
        o.update_coded(code_cols, dicts, n)                          # (and once more as part of the run below)
    st.update_coded_blocks(blocks, block_dicts)
    assert _factored_launches(capi) - before == 4, "misaligned stripes did not take the factored kernel"
    assert_same_groups(finalize_np(st, dev), o.finalize())


@pytest.mark.parametrize("pred_kind", ["plain_int", "coded_double", "two_terms", "date"])
def test_factored_aggregation_under_the_states_predicate(capi, oracle, dev, pred_kind, monkeypatch):
    """TPC-H Q1 as the reference runs it has its predicate inside the aggregation (l_shipdate <= DATE: the operator's
    predicate_, storage/AggregationOperationState.cpp:428-440).  A state that factors through the dictionary codes keeps doing
    so: a pass of its own (factored_predicate_kernel) evaluates the terms — on plain values, or on codes through the block's
    dictionary — into the filter bitmap of the call, ANDed with the caller's own filter.  One stripe and a run of blocks with
    per-block dictionaries, against the oracle (which evaluates the predicate row by row like the reference)."""
    monkeypatch.setenv("QSX_AGG_FACTORED_MIN_ROWS", "0")
    monkeypatch.setenv("QSX_AGG_JIT_MIN_ROWS", str(1 << 60))
    # terms on plain stripes go through the K1 kernels, chained through the bitmap; a term on a compressed attribute takes
    # factored_predicate_kernel (values through the dictionary), which a state only does when asked to
    if pred_kind in ("coded_double", "two_terms"):
        monkeypatch.setenv("QSX_AGG_FACTORED_CODED_PREDICATES", "1")
    else:
        monkeypatch.delenv("QSX_AGG_FACTORED_CODED_PREDICATES", raising=False)
    rng = np.random.default_rng(2100)
    spec = _FACTORED_PLANS["q1"]
    date_type = (T.DATE, None) if pred_kind == "date" else (T.INT, None)
    layout = [(T.CHAR, 1), (T.CHAR, 1)] + [(T.DOUBLE, None)] * 4 + [date_type]
    if pred_kind == "plain_int":
        pred = [(6, T.LE, 19980902)]
    elif pred_kind == "coded_double":
        pred = [(4, T.GE, 0.045)]                       # l_discount >= 0.045: a dictionary column, between two of its values
    elif pred_kind == "two_terms":
        pred = [(6, T.LE, 19980902), (2, T.LT, 24.0)]   # and l_quantity < 24 (a dictionary column)
    else:
        pred = [(6, T.LE, T.date_raw(1998, 9, 2))]
    from test_gpu_agg import assert_same_groups, finalize_np
    cfg = None
    st = o = None
    before = _factored_launches(capi)
    blocks, block_dicts, filters, host = [], [], [], []
    for b, n in enumerate((2048 * 4 + 9, 51_003, 700, 2048 * 3)):
        letters = np.frombuffer(b"ANR", dtype=np.uint8)
        qty = rng.choice(rng.choice(np.arange(1, 51), size=int(rng.integers(20, 51)), replace=False).astype(np.float64), size=n)
        disc = rng.choice(rng.choice(np.arange(0, 11), size=int(rng.integers(3, 12)), replace=False) / 100.0, size=n)
        if pred_kind == "date":
            years, months, days = rng.integers(1996, 2001, size=n), rng.integers(1, 13, size=n), rng.integers(1, 29, size=n)
            ship = ((years.astype(np.int64) & 0xFFFFFFFF) | (months.astype(np.int64) << 32) | (days.astype(np.int64) << 40) |
                    (rng.integers(0, 1 << 16, size=n).astype(np.int64) << 48)).astype(np.int64)     # (the padding bytes are never looked at)
        else:
            ship = (19980101 + rng.integers(0, 1200, size=n)).astype(np.int32)
        cols = [rng.choice(letters, size=n), rng.choice(np.frombuffer(b"FO", dtype=np.uint8), size=n), qty,
                np.round(rng.uniform(900, 105000, size=n), 2), disc, rng.integers(0, 9, size=n) / 100.0, ship]
        comp = {2: _coded(oracle, cols[2]), 4: _coded(oracle, cols[4]), 5: _coded(oracle, cols[5])}
        widths = [comp[i].code_width if i in comp else 0 for i in range(7)]
        if cfg is None:
            cfg = T.make_agg_config(T.AGG_COMPACT_KEY, layout, keys=[0, 1], instrs=spec["instrs"], consts=spec["consts"], aggs=spec["aggs"],
                                    est_groups=6, code_widths=widths, pred=pred)
            st, o = capi.AggState(cfg), oracle.AggState(cfg)
        code_cols = [comp[i].codes if i in comp else cols[i] for i in range(7)]
        dicts = [comp[i].dictionary if i in comp else None for i in range(7)]
        filt = oracle.bitmap_from_bools(rng.random(n) < 0.8) if b % 2 == 1 else None
        dev_cols = [to_dev(c, dev) for c in code_cols]
        dev_dicts = [None if d is None else to_dev(d, dev) for d in dicts]
        st.update_coded(dev_cols, dev_dicts, n, filter_bitmap=None if filt is None else bitmap_dev(filt, dev))     # one stripe
        o.update_coded(code_cols, dicts, n, filter_bitmap=filt)
        o.update_coded(code_cols, dicts, n, filter_bitmap=filt)                                                   # (and in the run below)
        blocks.append(dev_cols)
        block_dicts.append(dev_dicts)
        filters.append(None if filt is None else bitmap_dev(filt, dev))
        host.append((code_cols, dicts, n))
    st.update_coded_blocks(blocks, block_dicts, filters)
    st.update_coded_blocks(blocks[::2], block_dicts[::2])          # no filter of the caller's at all
    for i in (0, 2):
        o.update_coded(host[i][0], host[i][1], host[i][2])
    assert _factored_launches(capi) - before == 6, "a state with a predicate did not take the factored kernels"
    assert_same_groups(finalize_np(st, dev), o.finalize())


# ---------------------------------------------------------------------------------------------------------------------------
# Joins and LIP filters directly on compressed key stripes (qsx_*_blocks_coded, include/qsx.h)
# ---------------------------------------------------------------------------------------------------------------------------
def _key_blocks_of_every_kind(rng, dtype, rows, lo=0):
    """Blocks of one key attribute the way a CompressedColumnStore holds them: every block compressed on its own
    (oracle.CompressedColumn = CompressedBlockBuilder's decision), so that the run mixes dictionary codes of 1 and 2 bytes,
    truncated values of 1 / 2 / 4 bytes and plain values — and an empty block."""
    makers = [
        lambda n: rng.choice(rng.integers(lo, lo + 60_000, size=180), size=n),                    # few distinct values: 1-byte dictionary codes
        lambda n: rng.permutation(60_000)[:n] if n <= 60_000 else rng.integers(0, 60_000, size=n),  # < 2^16, many distinct: 2-byte truncation
        lambda n: rng.integers(-50, 90_000, size=n),                                              # negative values: never truncated
        lambda n: rng.choice(rng.integers(lo + 66_000, lo + 90_000, size=600), size=n),             # 600 distinct 17-bit values: 2-byte dictionary codes
        lambda n: rng.integers(0, 250, size=n),                                                   # < 2^8: 1-byte truncation (or dictionary)
        lambda n: rng.integers(70_000, 95_000, size=n),                                           # 17 bits: 4-byte truncation for a LONG
    ]
    blocks = []
    for i, n in enumerate(rows):
        values = makers[i % len(makers)](n).astype(dtype) if n else np.zeros(0, dtype=dtype)
        blocks.append(values)
    return blocks


def _coded_run(oracle, blocks, dev):
    """(device stripes as they lie, coding for the *_coded calls, the kinds seen)."""
    stripes, coding, kinds = [], [], set()
    for values in blocks:
        if values.size == 0:
            stripes.append(to_dev(values, dev))
            coding.append((0, None))
            continue
        col = oracle.CompressedColumn(values)
        assert np.array_equal(col.decode(), values)
        kinds.add((col.kind, col.code_width if col.kind else 0))
        if col.kind == 0:
            stripes.append(to_dev(values, dev))
            coding.append((0, None))
        else:
            stripes.append(codes_dev(col, dev))
            coding.append((col.code_width, None if col.dictionary is None else to_dev(col.dictionary, dev)))
    return stripes, coding, kinds


@pytest.mark.parametrize("flavour", ["hashed", "dense"])
@pytest.mark.parametrize("key_type,dtype", [(T.INT, np.int32), (T.LONG, np.int64)])
def test_joins_read_compressed_key_stripes_as_they_lie(capi, oracle, dev, flavour, key_type, dtype):
    """qsx_join_build_blocks_coded / qsx_join_probe_blocks_coded / _count_ / _exists_ / _project_: build and probe over runs of
    blocks whose key attribute lies as CompressedBlockBuilder left it (per block: values, truncated values, dictionary codes) —
    the reference reads such a key through CompressedTupleStorageSubBlock::getAttributeValue (storage/
    CompressedTupleStorageSubBlock.hpp:225-300) — give the pairs, counts, bitmaps and output tuples of the same calls over the
    decoded stripes, and of the oracle's join."""
    from test_gpu_join import sorted_pairs
    rng = np.random.default_rng(7 if flavour == "dense" else 8)
    build_rows = [5_000, 4_099, 7_000, 3_000, 2_500, 6_001, 0, 1_200]
    probe_rows = [9_000, 70_001, 20_000, 4_097, 33_000, 12_345, 0, 1, 777]
    build_blocks = _key_blocks_of_every_kind(rng, dtype, build_rows)
    probe_blocks = _key_blocks_of_every_kind(rng, dtype, probe_rows, lo=20_000)
    b_stripes, b_coding, b_kinds = _coded_run(oracle, build_blocks, dev)
    p_stripes, p_coding, p_kinds = _coded_run(oracle, probe_blocks, dev)
    want_kinds = {(0, 0), (1, 1), (1, 2), (2, 1), (2, 2)} | ({(1, 4)} if key_type == T.LONG else set())
    assert want_kinds <= (b_kinds | p_kinds), (b_kinds, p_kinds)                       # the run really mixes the forms
    n_build = sum(build_rows)
    key_range = (-50, 95_000) if flavour == "dense" else None
    bases = [int(x) for x in np.cumsum([0] + build_rows[:-1])]
    table = capi.JoinTable(key_type, n_build, key_range=key_range)
    table.build_blocks(b_stripes, bases, coding=b_coding)
    plain = capi.JoinTable(key_type, n_build, key_range=key_range)                      # the same joins over decoded stripes
    plain.build_blocks([to_dev(b, dev) for b in build_blocks], bases)
    ot = oracle.JoinTable(key_type, n_build)
    for b, blk in enumerate(build_blocks):
        ot.build(blk, block_id=0, base_tid=bases[b])
    assert table.size() == n_build
    d_probe = [to_dev(b, dev) for b in probe_blocks]
    filters = [oracle.bitmap_from_bools(rng.random(n) < 0.6) if (i % 3 != 1 and n) else None for i, n in enumerate(probe_rows)]
    dfilters = [None if f is None else bitmap_dev(f, dev) for f in filters]
    for use_filters in (False, True):
        f_arg = dfilters if use_filters else None
        want_p, want_b, start = [], [], 0
        for i, blk in enumerate(probe_blocks):
            rp, rb = ot.probe(blk, filter_bitmap=filters[i] if use_filters else None)
            want_p.append(rp + start)
            want_b.append(rb)
            start += blk.size
        want_p, want_b = np.concatenate(want_p), np.concatenate(want_b)
        for tbl, stripes, coding in ((table, p_stripes, p_coding), (table, d_probe, None), (plain, p_stripes, p_coding)):
            p, b, cnt = tbl.probe_blocks(stripes, capacity=want_p.size + 3, filters=f_arg, coding=coding)
            k = int(cnt.item())
            assert k == want_p.size
            assert np.array_equal(sorted_pairs(p.cpu().numpy()[:k], b.cpu().numpy()[:k]), sorted_pairs(want_p, want_b))
            assert int(tbl.probe_count_blocks(stripes, filters=f_arg, coding=coding).item()) == k
        for anti in (False, True):
            outs, cnt = table.probe_exists_blocks(p_stripes, anti=anti, filters=f_arg, coding=p_coding)
            total = 0
            for i, blk in enumerate(probe_blocks):
                ref = ot.probe_exists(blk, anti=anti, filter_bitmap=filters[i] if use_filters else None)
                if blk.size:
                    assert np.array_equal(bitmap_np(outs[i]), ref), (i, anti, use_filters)
                total += oracle.bitmap_count(ref, blk.size)
            assert int(cnt.item()) == total
    # code stripes at any byte address (a reference block image adopted in place, StorageManager::adoptBlockImage: the stripes
    # lie back to back behind the header, 2- and 4-byte codes at odd addresses)
    shifted = []
    for i, (stripe, (w, _)) in enumerate(zip(p_stripes, p_coding)):
        if w == 0 or stripe.numel() == 0:
            shifted.append(stripe)
            continue
        raw = stripe.view(torch.uint8)
        buf = torch.zeros(raw.numel() + 16, dtype=torch.uint8, device=dev)
        off = 1 + 2 * (i % 3)
        buf[off:off + raw.numel()] = raw
        shifted.append(buf[off:off + raw.numel()])
    p, b, cnt = table.probe_blocks(shifted, capacity=want_p.size + 3, filters=dfilters, coding=p_coding)
    assert int(cnt.item()) == want_p.size
    assert np.array_equal(sorted_pairs(p.cpu().numpy()[:want_p.size], b.cpu().numpy()[:want_p.size]), sorted_pairs(want_p, want_b))
    # the output relation written by the probe: the join key itself (given as the key stripes: emitted as its value), a probe
    # attribute and a build attribute.  Unique build keys here (one match per probe tuple at most).
    uniq = rng.permutation(95_000)[:n_build].astype(dtype)
    u_blocks = [np.sort(uniq[a:a + n]) if i % 2 else uniq[a:a + n] for i, (a, n) in enumerate(zip(bases, build_rows))]
    u_stripes, u_coding, _ = _coded_run(oracle, u_blocks, dev)
    utable = capi.JoinTable(key_type, n_build, key_range=(0, 95_000) if flavour == "dense" else None)
    utable.build_blocks(u_stripes, bases, coding=u_coding)
    payload = rng.integers(-2**40, 2**40, size=n_build).astype(np.int64)               # build attribute, by build tid
    probe_attr = [rng.integers(0, 1000, size=n).astype(np.int32) for n in probe_rows]
    try:
        outs, cnt = utable.probe_project_blocks(p_stripes, [p_stripes, [to_dev(a, dev) for a in probe_attr]], [[to_dev(payload, dev)]],
                                                coding=p_coding, key_dtype=torch.int32 if key_type == T.INT else torch.int64)
    except capi.QsxError as e:   # a hashed table that got no directly addressed shadow: the documented refusal
        assert flavour == "hashed" and e.status == T.ERR_UNSUPPORTED
        return
    k = int(cnt.item())
    all_keys = np.concatenate(probe_blocks)
    all_attr = np.concatenate(probe_attr)
    # (probe-side columns only — SELECT the join attribute of a semi-join-like inner join: the kernel without a covering array)
    outs2, cnt2 = utable.probe_project_blocks(p_stripes, [p_stripes, [to_dev(a, dev) for a in probe_attr]], [], coding=p_coding,
                                              key_dtype=torch.int32 if key_type == T.INT else torch.int64)
    k2 = int(cnt2.item())
    got2 = np.stack([outs2[0].cpu().numpy()[:k2].astype(np.int64), outs2[1].cpu().numpy()[:k2].astype(np.int64)], axis=1)
    where = {int(v): i for i, v in enumerate(np.concatenate(u_blocks))}
    hit = np.array([int(v) in where for v in all_keys])
    assert k == int(hit.sum())
    got = np.stack([outs[0].cpu().numpy()[:k].astype(np.int64), outs[1].cpu().numpy()[:k].astype(np.int64), outs[2].cpu().numpy()[:k]], axis=1)
    want = np.stack([all_keys[hit].astype(np.int64), all_attr[hit].astype(np.int64),
                     payload[[where[int(v)] for v in all_keys[hit]]]], axis=1)
    assert np.array_equal(got[np.lexsort(got.T[::-1])], want[np.lexsort(want.T[::-1])])
    assert k2 == k
    assert np.array_equal(got2[np.lexsort(got2.T[::-1])], want[:, :2][np.lexsort(want[:, :2].T[::-1])])


@pytest.mark.parametrize("key_type,dtype", [(T.INT, np.int32), (T.LONG, np.int64)])
def test_lip_filters_over_compressed_key_stripes(capi, oracle, dev, key_type, dtype):
    """qsx_lip_build_blocks_coded / qsx_lip_probe_blocks_coded equal the plain forms over the decoded stripes, bit for bit."""
    rng = np.random.default_rng(21)
    build_blocks = _key_blocks_of_every_kind(rng, dtype, [5_000, 3_000, 2_000, 900, 1_500, 800, 0, 3])
    probe_blocks = _key_blocks_of_every_kind(rng, dtype, [40_000, 8_193, 30_000, 2_000, 5_000, 6_000, 0, 1], lo=10_000)
    b_stripes, b_coding, _ = _coded_run(oracle, build_blocks, dev)
    p_stripes, p_coding, _ = _coded_run(oracle, probe_blocks, dev)
    for kind, cardinality in ((T.LIP_BITVECTOR_EXACT, 95_051), (T.LIP_SINGLE_IDENTITY_HASH, 4_001)):
        coded = capi.LipFilter(kind, cardinality, min_value=-50)
        plain = capi.LipFilter(kind, cardinality, min_value=-50)
        coded.build_blocks(b_stripes, coding=b_coding, key_type=key_type)
        plain.build_blocks([to_dev(b, dev) for b in build_blocks if b.size])
        assert torch.equal(coded.export(dev), plain.export(dev))
        in_bitmaps = [bitmap_dev(oracle.bitmap_from_bools(rng.random(b.size) < 0.7), dev) if (i % 2 and b.size) else None
                      for i, b in enumerate(probe_blocks)]
        for ins in (None, in_bitmaps):
            outs, cnt = coded.probe_blocks(p_stripes, in_bitmaps=ins, coding=p_coding, key_type=key_type)
            ref, ref_cnt = plain.probe_blocks([to_dev(b, dev) for b in probe_blocks], in_bitmaps=ins)
            assert int(cnt.item()) == int(ref_cnt.item())
            for i, b in enumerate(probe_blocks):
                if b.size:
                    assert torch.equal(outs[i], ref[i]), (kind, i)


@pytest.mark.parametrize("key_range", [None, (0, 4_999)])
def test_small_build_side_probed_with_sorted_truncated_keys(capi, oracle, dev, key_range):
    """A dimension of a few thousand keys (its directly addressed form fits LDS) probed with runs of blocks sorted on the join
    attribute and truncated to 2 bytes — tests/cpp/block_image_test.cpp's join over adopted CompressedColumnStore images — through
    every coded probe form; the projection is the join attribute alone."""
    rng = np.random.default_rng(5)
    dim = np.arange(0, 5_000, 3, dtype=np.int32)
    table = capi.JoinTable(T.INT, dim.size, key_range=key_range)
    table.build(to_dev(dim, dev))
    rows = [120_000, 118_991, 0, 116_973, 40_001]
    blocks = [np.sort(rng.integers(0, 5_000, size=n)).astype(np.int32) for n in rows]
    stripes, coding, kinds = _coded_run(oracle, blocks, dev)
    assert kinds == {(1, 2)}
    want = int(sum((b % 3 == 0).sum() for b in blocks))
    assert int(table.probe_count_blocks(stripes, coding=coding).item()) == want
    p, b, cnt = table.probe_blocks(stripes, coding=coding)
    assert int(cnt.item()) == want
    all_keys = np.concatenate(blocks)
    pk = p.cpu().numpy()[:want]
    assert np.array_equal(np.sort(all_keys[pk]), np.sort(all_keys[all_keys % 3 == 0]))
    assert np.array_equal(dim[b.cpu().numpy()[:want]], all_keys[pk])
    outs, cnt = table.probe_exists_blocks(stripes, coding=coding)
    assert int(cnt.item()) == want
    try:
        outs, cnt = table.probe_project_blocks(stripes, [stripes], [], coding=coding, key_dtype=torch.int32)
    except capi.QsxError as e:
        assert e.status == T.ERR_UNSUPPORTED and key_range is None
        outs, cnt = table.probe_project_blocks(stripes, [[to_dev(x, dev) for x in blocks]], [], coding=coding, key_dtype=torch.int32)
    assert int(cnt.item()) == want
    assert np.array_equal(np.sort(outs[0].cpu().numpy()[:want]), np.sort(all_keys[all_keys % 3 == 0]))


@pytest.mark.parametrize("types,dtypes", [((T.INT, T.INT), (np.int32, np.int32)), ((T.INT, T.LONG, T.INT), (np.int32, np.int64, np.int32))])
def test_composite_keys_packed_from_compressed_components(capi, oracle, dev, types, dtypes):
    """qsx_join_key_pack_blocks_coded: the packed (exact, <= 64 bits) or folded (wider) composite key of a run of blocks whose
    components lie compressed per block equals qsx_join_key_pack_blocks over the decoded stripes, bit for bit."""
    rows = [5_000, 1, 0, 7_003, 2_500, 4_097, 513]
    comps = [_key_blocks_of_every_kind(np.random.default_rng(11 + k), dt, rows, lo=1000 * k) for k, dt in enumerate(dtypes)]
    blocks, coding, plain = [], [], []
    for b in range(len(rows)):
        stripes, cods, vals = [], [], []
        for k in range(len(types)):
            values = comps[k][(b + 2 * k) % len(rows)]
            values = np.resize(values, rows[b]).astype(dtypes[k]) if rows[b] else np.zeros(0, dtype=dtypes[k])
            (s,), (c,), _ = _coded_run(oracle, [values], dev)
            stripes.append(s)
            cods.append(c)
            vals.append(to_dev(values, dev))
        blocks.append(stripes)
        coding.append(cods)
        plain.append(vals)
    got, exact = capi.join_key_pack_blocks_coded(blocks, coding, list(types))
    want, want_exact = capi.join_key_pack_blocks([p for p in plain])
    assert exact == want_exact == (sum(32 if t == T.INT else 64 for t in types) <= 64)
    assert torch.equal(got, want)
    assert any(c[0] != 0 for cs in coding for c in cs), "no component of any block was compressed"
