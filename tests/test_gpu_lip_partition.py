"""GPU parity: K12 LIP filters and K9 hash-partition scatter against the oracle."""
import numpy as np
import pytest

from quickstep_amd import types as T
from helpers import bitmap_dev, bitmap_np, to_dev

pytestmark = pytest.mark.gpu


def test_golden_lip_semi_join(capi, oracle, dev, golden):
    lip = golden["sql_golden"]["lip"]
    x = np.arange(0, lip["limit"] + 1, lip["r_step"], dtype=np.int32)
    z = np.arange(0, lip["limit"] + 1, lip["s_step"], dtype=np.int32)
    f = capi.LipFilter(T.LIP_BITVECTOR_EXACT, int(z.max() - z.min() + 1), int(z.min()))
    f.build(to_dev(z, dev))
    bm, cnt = f.probe(to_dev(x, dev))
    hit = oracle.bools_from_bitmap(bitmap_np(bm), x.size)
    semi = x[hit]
    assert semi[semi % 10000 == 0].tolist() == lip["semi_join_mod_10000"]
    assert int(semi[semi % 5 == 0].sum()) + int(semi[semi % 7 == 0].sum()) == lip["sum_x_union_mod5_mod7"]
    assert int(cnt.item()) == semi.size


@pytest.mark.parametrize("dtype", [np.int32, np.int64])
@pytest.mark.parametrize("kind,anti", [(T.LIP_SINGLE_IDENTITY_HASH, False), (T.LIP_BITVECTOR_EXACT, False),
                                       (T.LIP_BITVECTOR_EXACT, True)])
def test_lip_filters_match_oracle(capi, oracle, dev, dtype, kind, anti):
    rng = np.random.default_rng(4)
    build = rng.integers(100, 5000, size=3000).astype(dtype)
    probe = rng.integers(-200, 6000, size=100_001).astype(dtype)       # negative / out-of-range keys on the probe side
    card, mn = (4099, 0) if kind == T.LIP_SINGLE_IDENTITY_HASH else (int(build.max() - build.min() + 1), int(build.min()))
    bfilt = oracle.bitmap_from_bools(rng.random(build.size) < 0.9)
    pin = oracle.bitmap_from_bools(rng.random(probe.size) < 0.6)
    f = capi.LipFilter(kind, card, mn, anti)
    o = oracle.LipFilter(kind, card, mn, anti)
    f.build(to_dev(build, dev), filter_bitmap=bitmap_dev(bfilt, dev))
    o.build(build, filter_bitmap=bfilt)
    for inb in (None, pin):
        bm, cnt = f.probe(to_dev(probe, dev), in_bitmap=None if inb is None else bitmap_dev(inb, dev))
        ref = o.probe(probe, in_bitmap=inb)
        assert np.array_equal(bitmap_np(bm), ref)
        assert int(cnt.item()) == oracle.bitmap_count(ref, probe.size)


def test_golden_partition_membership(capi, dev, golden):
    g = golden["hash_partition"]["partition_by_hash_4"]
    keys = np.array(g["ids_inserted"], dtype=np.int32)
    (out,), offs = capi.partition_scatter(to_dev(keys, dev), 4, [to_dev(keys, dev)])
    offs = offs.cpu().numpy()
    out = out.cpu().numpy()
    assert [out[offs[p]:offs[p + 1]].tolist() for p in range(4)] == g["expected_partitions"]


@pytest.mark.parametrize("dtype", [np.int32, np.int64])
@pytest.mark.parametrize("P", [1, 2, 3, 8, 41, 64])
@pytest.mark.parametrize("n", [0, 1, 1000, 777_777])
def test_partition_scatter_is_stable_and_matches_oracle(capi, oracle, dev, dtype, P, n):
    rng = np.random.default_rng(n + P)
    keys = rng.integers(-1000, 1_000_000, size=n).astype(dtype)
    payload8 = rng.integers(0, 2**62, size=n).astype(np.int64)
    payload1 = rng.integers(0, 255, size=n).astype(np.uint8)
    outs, offs = capi.partition_scatter(to_dev(keys, dev), P, [to_dev(keys, dev), to_dev(payload8, dev), to_dev(payload1, dev)])
    assert np.array_equal(offs.cpu().numpy(), oracle.partition_offsets(keys, P))
    for got, col in zip(outs, (keys, payload8, payload1)):
        assert np.array_equal(got.cpu().numpy(), oracle.partition_scatter(keys, P, col) if n else col)
