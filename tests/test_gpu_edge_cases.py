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
