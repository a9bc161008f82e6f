"""GPU parity for compressed attributes: code-stripe scans and decode through the C ABI
against the oracle's restatement of CompressedColumnStoreTupleStorageSubBlock / CompressedStoreUtil."""
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
