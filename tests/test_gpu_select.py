"""GPU parity: K1/K2/K5 through the C ABI against the oracle (bit-exact)."""
import numpy as np
import pytest
import torch

from quickstep_amd import types as T
from helpers import bitmap_dev, bitmap_np, to_dev

pytestmark = pytest.mark.gpu

SIZES = [0, 1, 63, 64, 65, 4095, 4096, 4097, 100_003, 1_000_000]
DTYPES = [np.int32, np.int64, np.float32, np.float64]


def make_col(rng, dtype, n):
    if np.issubdtype(dtype, np.integer):
        return rng.integers(-50, 50, size=n).astype(dtype)
    col = rng.integers(-50, 50, size=n).astype(dtype) / 4
    if n > 10:
        col[rng.integers(0, n, size=3)] = np.nan     # NaN compares false except !=
    return col


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n", SIZES)
def test_select_cmp_bitmaps_are_bit_exact(capi, oracle, dev, dtype, n):
    rng = np.random.default_rng(n + 17)
    col = make_col(rng, dtype, n)
    dcol = to_dev(col, dev)
    for op in (T.EQ, T.NE, T.LT, T.LE, T.GT, T.GE):
        bm, cnt = capi.select_cmp(dcol, op, 3)
        ref = oracle.select_cmp(col, op, 3)
        if n:
            assert np.array_equal(bitmap_np(bm)[:ref.size], ref), (dtype, n, op)
        assert int(cnt.item()) == oracle.bitmap_count(ref, n)


@pytest.mark.parametrize("n", [1, 64, 1000, 250_001])
def test_select_cmp_with_filter_and_bitmap_algebra(capi, oracle, dev, n):
    rng = np.random.default_rng(n)
    col = rng.integers(0, 1000, size=n).astype(np.int32)
    dcol = to_dev(col, dev)
    first, _ = capi.select_cmp(dcol, T.GE, 100)
    second, cnt = capi.select_cmp(dcol, T.LT, 600, filter_bitmap=first)
    rfirst = oracle.select_cmp(col, T.GE, 100)
    rsecond = oracle.select_cmp(col, T.LT, 600, filter_bitmap=rfirst)
    assert np.array_equal(bitmap_np(second), rsecond)
    assert int(cnt.item()) == int(((col >= 100) & (col < 600)).sum())
    other, _ = capi.select_cmp(dcol, T.LT, 300)
    a, b = oracle.bools_from_bitmap(rfirst, n), col < 300
    for op, want in ((0, a & b), (1, a | b), (2, a & ~b), (3, ~a)):
        got = capi.bitmap_combine(op, first, other, n)
        assert np.array_equal(bitmap_np(got), oracle.bitmap_from_bools(want)), op   # trailing bits stay zero
        assert int(capi.bitmap_count(got, n).item()) == int(want.sum())


def test_golden_bitvector_words(capi, dev, golden):
    for case in golden["bitvector"]["cases"]:
        col = np.zeros(case["n"], dtype=np.int32)
        col[case["set_bits"]] = 1
        bm, cnt = capi.select_cmp(to_dev(col, dev), T.EQ, 1)
        assert [f"{w:016x}" for w in bitmap_np(bm)] == case["expected_words_hex"]
        tids, c = capi.bitmap_to_tids(bm, case["n"])
        assert tids.cpu().numpy()[:int(c.item())].tolist() == case["set_bits"]


@pytest.mark.parametrize("n", [0, 1, 64, 4097, 300_007])
@pytest.mark.parametrize("selectivity", [0.0, 0.01, 0.5, 1.0])
def test_compact_gather_is_order_preserving(capi, oracle, dev, n, selectivity):
    rng = np.random.default_rng(n + int(selectivity * 100))
    keep = rng.random(n) < selectivity
    words = oracle.bitmap_from_bools(keep) if n else np.zeros(1, dtype=np.uint64)
    cols = [rng.integers(0, 255, size=n).astype(np.uint8), rng.integers(-2**15, 2**15, size=n).astype(np.int16),
            rng.integers(-2**31, 2**31, size=n).astype(np.int32), rng.normal(size=n)]
    out, cnt = capi.compact_gather([to_dev(c, dev) for c in cols], bitmap_dev(words, dev), n)
    k = int(cnt.item())
    assert k == int(keep.sum())
    for o, c in zip(out, cols):
        assert np.array_equal(o.cpu().numpy()[:k], oracle.compact_gather(c, words) if n else c[:0])
    tids, c2 = capi.bitmap_to_tids(bitmap_dev(words, dev), n, base_tid=7)
    assert np.array_equal(tids.cpu().numpy()[:int(c2.item())], np.nonzero(keep)[0].astype(np.int32) + 7)


def test_gather_by_tuple_id(capi, oracle, dev):
    rng = np.random.default_rng(2)
    for dtype in (np.uint8, np.int16, np.int32, np.float64):
        src = rng.integers(0, 100, size=5000).astype(dtype)
        tids = rng.integers(-1, 5000, size=20_001).astype(np.int32)      # -1 = outer-join NULL padding
        got = capi.gather(to_dev(src, dev), to_dev(tids, dev))
        assert np.array_equal(got.cpu().numpy(), oracle.gather(src, tids))


def test_select_config_c1_selectivities(capi, oracle, dev):
    """BASELINE config 1 shape at 2 M rows: col < K for ~1 %, 10 %, 50 % then project col."""
    rng = np.random.default_rng(1)
    col = rng.integers(0, 2**31, size=2_000_000).astype(np.int32)
    dcol = to_dev(col, dev)
    for k in (21474836, 214748364, 1073741824):
        bm, cnt = capi.select_cmp(dcol, T.LT, k)
        (out,), c = capi.compact_gather([dcol], bm, col.size)
        assert int(c.item()) == int(cnt.item()) == int((col < k).sum())
        assert np.array_equal(out.cpu().numpy()[:int(c.item())], col[col < k])


@pytest.mark.parametrize("dtype", [np.int32, np.int64, np.float32, np.float64])
@pytest.mark.parametrize("n", [1, 63, 64, 65, 100_003])
def test_select_cmp_columns_is_bit_exact(capi, oracle, dev, dtype, n):
    """attribute OP attribute (LiteralComparators-inl.hpp:52-125), with and without a filter."""
    rng = np.random.default_rng(n)
    a = rng.integers(-5, 5, size=n).astype(dtype)
    b = rng.integers(-5, 5, size=n).astype(dtype)
    f = oracle.bitmap_from_bools(rng.random(n) < 0.6)
    da, db = to_dev(a, dev), to_dev(b, dev)
    for op in range(6):
        bm, cnt = capi.select_cmp_columns(da, db, op)
        want = oracle.select_cmp_columns(a, b, op)
        assert np.array_equal(bitmap_np(bm), want)
        assert int(cnt.item()) == oracle.bitmap_count(want, n)
        bm, cnt = capi.select_cmp_columns(da, db, op, filter_bitmap=bitmap_dev(f, dev))
        want = oracle.select_cmp_columns(a, b, op, filter_bitmap=f)
        assert np.array_equal(bitmap_np(bm), want)
        assert int(cnt.item()) == oracle.bitmap_count(want, n)


def test_tids_to_bitmap_round_trip(capi, oracle, dev):
    rng = np.random.default_rng(12)
    n_bits = 70_001
    tids = rng.integers(1000, 1000 + n_bits, size=200_000).astype(np.int32)     # duplicates on purpose
    tids[::1000] = 5                                                            # below base: ignored
    got = bitmap_np(capi.tids_to_bitmap(to_dev(tids, dev), n_bits, base_tid=1000))
    want = oracle.tids_to_bitmap(tids, n_bits, base_tid=1000)
    assert np.array_equal(got, want)
    back, cnt = capi.bitmap_to_tids(capi.tids_to_bitmap(to_dev(tids, dev), n_bits, base_tid=1000), n_bits, base_tid=1000)
    uniq = np.unique(tids[tids >= 1000])
    assert np.array_equal(back[:int(cnt.item())].cpu().numpy(), uniq)
    empty = capi.tids_to_bitmap(torch.empty(0, dtype=torch.int32, device=dev), 130)
    assert not bitmap_np(empty).any()


@pytest.mark.parametrize("dtype", [np.int32, np.int64, np.float32, np.float64])
def test_select_cmp_aligned_and_unaligned_stripes_agree(capi, oracle, dev, dtype):
    """16-byte aligned stripes take the packed kernel (16 bytes per lane per load), slices that start inside a 16-byte
    chunk the row-per-lane kernel; both must give the oracle's words for every offset and ragged length."""
    rng = np.random.default_rng(8)
    base = rng.integers(-20, 20, size=70_000).astype(dtype)
    dbase = to_dev(base, dev)
    f_all = rng.random(70_000) < 0.7
    for start in (0, 1, 2, 3, 4, 5, 7, 8):
        for n in (1, 15, 16, 17, 63, 64, 65, 1023, 1025, 4097, 33_333):
            sl, dsl = base[start:start + n], dbase[start:start + n]
            f = oracle.bitmap_from_bools(f_all[:n])
            for op in (T.LT, T.EQ, T.GE):
                bm, cnt = capi.select_cmp(dsl, op, 3)
                want = oracle.select_cmp(sl, op, dtype(3))
                assert np.array_equal(bitmap_np(bm), want), (start, n, op)
                assert int(cnt.item()) == oracle.bitmap_count(want, n)
                bm, cnt = capi.select_cmp(dsl, op, 3, filter_bitmap=bitmap_dev(f, dev))
                want = oracle.select_cmp(sl, op, dtype(3), filter_bitmap=f)
                assert np.array_equal(bitmap_np(bm), want), (start, n, op, "filter")


@pytest.mark.parametrize("n", [1, 63, 64, 65, 4097, 1_000_003])
def test_sort_column_predicate_by_binary_search(capi, oracle, dev, n):
    """qsx_select_cmp_sorted on a sorted stripe = the scan's answer (qsx_select_cmp) = the oracle's lower/upper bound
    restatement of SortColumnPredicateEvaluator, for every comparison, type, literal position and with a filter."""
    rng = np.random.default_rng(n)
    filt = oracle.bitmap_from_bools(rng.random(n) < 0.7)
    cols = {
        "int32 with ties": np.sort(rng.integers(-50, 50, size=n)).astype(np.int32),
        "int64": np.sort(rng.integers(-2**40, 2**40, size=n)).astype(np.int64),
        "float32": np.sort((rng.normal(size=n) * 100).astype(np.float32)),
        "float64 ties": np.sort(np.round(rng.normal(size=n), 1)),
    }
    for name, col in cols.items():
        d = to_dev(col, dev)
        lits = [col[0], col[-1], col[n // 2], col[0] - 1, col[-1] + 1, (col[n // 3] + col[2 * n // 3]) / 2]
        for lit in lits:
            lit = col.dtype.type(lit).item()
            for op in range(6):
                for f in (None, filt):
                    got, cnt = capi.select_cmp_sorted(d, op, lit, filter_bitmap=None if f is None else bitmap_dev(f, dev))
                    want = oracle.select_cmp_sorted(col, op, lit, f)
                    assert np.array_equal(bitmap_np(got), want), (name, op, lit)
                    assert np.array_equal(want, oracle.select_cmp(col, op, lit, f))       # the scan agrees
                    assert int(cnt.item()) == oracle.bitmap_count(want, n)


# ---- DATE and CHAR(n) predicates (the predicate types of TPC-H Q1 / Q3 beside the numeric ones) ----------------------------
def make_dates(rng, n, padding=True):
    """Raw DateLit bytes as int64: years around the TPC-H range plus a few far-away ones, garbage in the padding bytes."""
    years = rng.integers(1992, 1999, size=n)
    if n > 10:
        years[rng.integers(0, n, size=3)] = [-18017, 99999, -99999]
    months = rng.integers(1, 13, size=n)
    days = rng.integers(1, 29, size=n)
    pad = rng.integers(0, 1 << 16, size=n) if padding else np.zeros(n, dtype=np.int64)
    raw = (years.astype(np.int64) & 0xFFFFFFFF) | (months.astype(np.int64) << 32) | (days.astype(np.int64) << 40) | (pad.astype(np.int64) << 48)
    return raw.astype(np.int64)


@pytest.mark.parametrize("n", [0, 1, 65, 4097, 300_001])
def test_date_predicates_are_bit_exact(capi, oracle, dev, n, golden):
    rng = np.random.default_rng(n + 5)
    col = make_dates(rng, n)
    dcol = to_dev(col, dev)
    keep = oracle.bitmap_from_bools(rng.random(n) < 0.6) if n else None
    for y, m, d in [(1995, 3, 15), (1998, 9, 2), (1992, 1, 1)] + [tuple(x) for x in golden["comparison_unittest"]["dates"]]:
        lit = T.date_raw(y, m, d)
        for op in (T.EQ, T.NE, T.LT, T.LE, T.GT, T.GE):
            for filt in (None, keep):
                bm, cnt = capi.select_cmp(dcol, op, lit, qtype=T.DATE, filter_bitmap=None if filt is None else bitmap_dev(filt, dev))
                ref = oracle.select_cmp(col, op, lit, filter_bitmap=filt, qt=T.DATE)
                if n:
                    assert np.array_equal(bitmap_np(bm)[:ref.size], ref), (n, op, y, m, d)
                assert int(cnt.item()) == oracle.bitmap_count(ref, n)
    if n:
        other = make_dates(rng, n)
        for op in (T.EQ, T.LT, T.GE):
            bm, cnt = capi.select_cmp_columns(dcol, to_dev(other, dev), op, qtype=T.DATE)
            ref = oracle.select_cmp_columns(col, other, op, qt=T.DATE)
            assert np.array_equal(bitmap_np(bm)[:ref.size], ref)
        # the sort-column path: order the stripe by (year, month, day) first
        order = oracle.sort_permutation([col], types=[T.DATE])
        sorted_col = np.ascontiguousarray(col[order])
        for op in (T.EQ, T.NE, T.LT, T.LE, T.GT, T.GE):
            lit = int(sorted_col[n // 2])
            bm, cnt = capi.select_cmp_sorted(to_dev(sorted_col, dev), op, lit, qtype=T.DATE)
            ref = oracle.select_cmp(sorted_col, op, lit, qt=T.DATE)
            assert np.array_equal(bitmap_np(bm)[:ref.size], ref)
            assert np.array_equal(oracle.select_cmp_sorted(sorted_col, op, lit, qt=T.DATE), ref)


@pytest.mark.parametrize("width,n", [(1, 1000), (10, 0), (10, 1), (10, 150_001), (25, 70_000), (15, 4096), (255, 5_000), (7, 64)])
def test_char_predicates_follow_strcmp_helper(capi, oracle, dev, width, n):
    """CHAR(width) OP literal: strings end at the first NUL or at the field width, unsigned bytes, a prefix is smaller
    (AsciiStringComparators.hpp:218-251).  TPC-H shaped values (c_mktsegment CHAR(10), l_shipinstruct CHAR(25), ...), values
    that fill the field exactly, bytes >= 0x80, an unaligned slice of the stripe, a filter."""
    rng = np.random.default_rng(width * 1000 + n)
    words = [b"BUILDING", b"AUTOMOBILE", b"MACHINERY", b"HOUSEHOLD", b"FURNITURE", b"BUILD", b"BUILDINGS", b"", b"\xc3\xa9t\xc3\xa9",
             b"DELIVER IN PERSON", b"TAKE BACK RETURN", b"COLLECT COD", b"NONE", b"x" * width]
    col = np.zeros((n, width), dtype=np.uint8)
    pick = rng.integers(0, len(words), size=n)
    for i in range(n):
        w = words[pick[i]][:width]
        col[i, :len(w)] = np.frombuffer(w, dtype=np.uint8)
        if len(w) + 1 < width and rng.random() < 0.2:
            col[i, len(w) + 1:] = rng.integers(1, 255, size=width - len(w) - 1)   # bytes behind the terminator are not part of the string
    dcol = to_dev(col, dev)
    keep = oracle.bitmap_from_bools(rng.random(n) < 0.5) if n else None
    literals = [b"BUILDING", b"BUILD", b"BUILDINGS", b"", b"MACHINERY", b"NONE\0junk", b"x" * min(width, 64), b"\xc3\xa9t\xc3\xa9", b"TAKE BACK RETURN"]
    for lit in literals:
        for op in (T.EQ, T.NE, T.LT, T.LE, T.GT, T.GE):
            for filt in (None, keep):
                bm, cnt = capi.select_cmp_char(dcol, op, lit, filter_bitmap=None if filt is None else bitmap_dev(filt, dev))
                ref = oracle.select_cmp_char(col, op, lit, filter_bitmap=filt)
                if n:
                    assert np.array_equal(bitmap_np(bm)[:ref.size], ref), (width, n, lit, op)
                assert int(cnt.item()) == oracle.bitmap_count(ref, n)
    if n > 200:
        # a slice that starts at row 3: not 16-byte aligned for most widths
        sl = dcol[3:3 + (n - 3) // 64 * 64]
        bm, cnt = capi.select_cmp_char(sl, T.EQ, b"BUILDING")
        ref = oracle.select_cmp_char(np.ascontiguousarray(col[3:3 + (n - 3) // 64 * 64]), T.EQ, b"BUILDING")
        assert np.array_equal(bitmap_np(bm)[:ref.size], ref)
    with pytest.raises(capi.QsxError):
        capi.select_cmp_char(to_dev(np.zeros((4, 10), dtype=np.uint8), dev), T.EQ, b"y" * 65)     # literal beyond QSX_MAX_CHAR_LITERAL


RUN_SHAPES = {
    "ragged": [5000, 0, 1, 63, 64, 65, 1023, 1024, 1025, 4096, 100_003, 0, 17],
    "equal": [20_000] * 9,
    "equal_short_last": [8192] * 6 + [100],
    "one": [250_001],
    "empty_only": [0, 0],
}


@pytest.mark.parametrize("shape", sorted(RUN_SHAPES))
@pytest.mark.parametrize("dtype", DTYPES)
def test_select_cmp_blocks_equals_block_by_block(capi, oracle, dev, dtype, shape):
    """One launch over a run of blocks (every block its own stripe, filter and bitmap) = the oracle block by block."""
    rows = RUN_SHAPES[shape]
    rng = np.random.default_rng(len(rows) + rows[0])
    cols = [make_col(rng, dtype, n) for n in rows]
    dcols = [to_dev(c, dev) for c in cols]
    filters = [oracle.bitmap_from_bools(rng.random(n) < 0.6) if (i % 3 != 1 and n) else None for i, n in enumerate(rows)]
    dfilters = [None if f is None else bitmap_dev(f, dev) for f in filters]
    for op in (T.EQ, T.NE, T.LT, T.LE, T.GT, T.GE):
        for use_filters in (False, True):
            outs, counts = capi.select_cmp_blocks(dcols, op, 3, filters=dfilters if use_filters else None)
            for b, n in enumerate(rows):
                ref = oracle.select_cmp(cols[b], op, dtype(3), filter_bitmap=filters[b] if use_filters else None)
                if n:
                    assert np.array_equal(bitmap_np(outs[b])[:ref.size], ref), (shape, b, n, op, use_filters)
                assert int(counts[b].item()) == oracle.bitmap_count(ref, n), (shape, b, op)


def test_select_cmp_blocks_unaligned_stripes_and_dates(capi, oracle, dev):
    rng = np.random.default_rng(21)
    base = rng.integers(-20, 20, size=50_000).astype(np.int32)
    dbase = to_dev(base, dev)
    cuts = [(0, 4000), (4001, 9000), (9003, 9003), (9003, 30_001), (30_002, 50_000)]     # stripes starting inside a 16-byte chunk
    outs, counts = capi.select_cmp_blocks([dbase[a:b] for a, b in cuts], T.LT, 3)
    for i, (a, b) in enumerate(cuts):
        ref = oracle.select_cmp(base[a:b], T.LT, np.int32(3))
        if b > a:
            assert np.array_equal(bitmap_np(outs[i])[:ref.size], ref)
        assert int(counts[i].item()) == oracle.bitmap_count(ref, b - a)
    dates = [make_dates(rng, n) for n in (3000, 70_001, 5)]
    lit = T.date_raw(1995, 3, 15)
    for op in (T.EQ, T.LT, T.GE):
        outs, counts = capi.select_cmp_blocks([to_dev(d, dev) for d in dates], op, lit, qtype=T.DATE)
        for i, d in enumerate(dates):
            ref = oracle.select_cmp(d, op, lit, qt=T.DATE)
            assert np.array_equal(bitmap_np(outs[i])[:ref.size], ref)
            assert int(counts[i].item()) == oracle.bitmap_count(ref, d.size)


@pytest.mark.parametrize("shape", sorted(RUN_SHAPES))
@pytest.mark.parametrize("selectivity", [0.0, 0.02, 0.6, 1.0])
def test_compact_gather_blocks_concatenates_in_block_and_row_order(capi, oracle, dev, shape, selectivity):
    """K2 over a run: the selected rows of every block, block after block, as the oracle's per-block compaction gives them."""
    rows = RUN_SHAPES[shape]
    rng = np.random.default_rng(len(rows) + int(selectivity * 50))
    blocks, keeps = [], []
    for n in rows:
        blocks.append([rng.integers(0, 255, size=n).astype(np.uint8), rng.integers(-2**15, 2**15, size=n).astype(np.int16),
                       rng.integers(-2**31, 2**31, size=n).astype(np.int32), rng.normal(size=n)])
        keeps.append(rng.random(n) < selectivity)
    words = [oracle.bitmap_from_bools(k) if k.size else np.zeros(1, dtype=np.uint64) for k in keeps]
    dblocks = [[to_dev(c, dev) for c in b] for b in blocks]
    dwords = [bitmap_dev(w, dev) for w in words]
    bases = [int(x) for x in np.cumsum([0] + rows[:-1]) + 100 * np.arange(len(rows))]
    for base_tids in (None, bases):
        out, tids, cnt = capi.compact_gather_blocks(dblocks, dwords, base_tids=base_tids, want_tids=True)
        k = int(cnt.item())
        assert k == sum(int(x.sum()) for x in keeps)
        for c in range(4):
            want = np.concatenate([oracle.compact_gather(blocks[b][c], words[b]) if rows[b] else blocks[b][c][:0] for b in range(len(rows))])
            assert np.array_equal(out[c].cpu().numpy()[:k], want), (shape, c)
        start, want_tids = 0, []
        for b, n in enumerate(rows):
            want_tids.append(np.nonzero(keeps[b])[0] + (start if base_tids is None else base_tids[b]))
            start += n
        assert np.array_equal(tids.cpu().numpy()[:k], np.concatenate(want_tids).astype(np.int32))


@pytest.mark.parametrize("width", [1, 10, 25, 255])
def test_char_predicates_over_runs_of_blocks(capi, oracle, dev, width):
    """qsx_select_cmp_char_blocks: one launch over the CHAR(width) stripes of a run of blocks = the comparison block by block."""
    rng = np.random.default_rng(width)
    words = [b"BUILDING", b"AUTOMOBILE", b"MACHINERY", b"BUILD", b"BUILDINGS", b"", b"x" * width]
    rows = [3000, 0, 1, 63, 64, 65, 1024, 1025, 20_001, 0, 33]
    blocks = []
    for n in rows:
        col = np.zeros((n, width), dtype=np.uint8)
        pick = rng.integers(0, len(words), size=n)
        for i in range(n):
            w = words[pick[i]][:width]
            col[i, :len(w)] = np.frombuffer(w, dtype=np.uint8)
        blocks.append(col)
    dblocks = [to_dev(b, dev) for b in blocks]
    filters = [oracle.bitmap_from_bools(rng.random(n) < 0.5) if (i % 3 != 1 and n) else None for i, n in enumerate(rows)]
    for lit in (b"BUILDING", b"BUILD", b"", b"x" * min(width, 64)):
        for op in (T.EQ, T.NE, T.LT, T.GE):
            for use_filters in (False, True):
                outs, counts = capi.select_cmp_char_blocks(dblocks, op, lit,
                                                           filters=[None if f is None else bitmap_dev(f, dev) for f in filters] if use_filters else None)
                for b, n in enumerate(rows):
                    ref = oracle.select_cmp_char(blocks[b], op, lit, filter_bitmap=filters[b] if use_filters else None)
                    if n:
                        assert np.array_equal(bitmap_np(outs[b])[:ref.size], ref), (width, b, n, lit, op)
                    assert int(counts[b].item()) == oracle.bitmap_count(ref, n)
