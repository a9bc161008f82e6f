"""GPU parity: K12 LIP filters and K9 hash-partition scatter against the oracle."""
import numpy as np
import pytest
import torch

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
def test_exact_lip_filter_read_off_a_directly_addressed_join_table(capi, oracle, dev, dtype):
    """qsx_lip_build_from_join_table: the filter of a BuildHash work order is built from the keys its table was built from, and a
    directly addressed table is an existence map of its key range — the filter read off the table's head words has the words of
    the filter built key by key (and of the oracle's).  Filters whose range starts before / inside / at the table's (offsets that
    are no multiple of 64), narrower and wider than it; duplicate keys (overflow chains), a build under a filter bitmap, two
    builds one after the other; a hashed table, an identity-hash filter and a run too short to pay are declined (nothing set)."""
    rng = np.random.default_rng(21)
    lo, hi = 1000, 1000 + 700_000
    keys = rng.integers(lo, hi, size=200_000).astype(dtype)                  # duplicates among them
    keep = oracle.bitmap_from_bools(rng.random(keys.size) < 0.7)
    more = rng.integers(lo, hi, size=50_000).astype(dtype)
    table = capi.JoinTable(T.INT if dtype == np.int32 else T.LONG, keys.size + more.size, key_range=(lo, hi - 1))
    table.build(to_dev(keys, dev), filter_bitmap=bitmap_dev(keep, dev))
    table.build(to_dev(more, dev), base_tid=keys.size)
    probe = np.arange(lo - 300, hi + 300).astype(dtype)
    for mn, card in ((lo, hi - lo), (lo - 131, hi - lo + 500), (lo + 77, 300_000), (lo - 64, 64 * 5000), (lo + 640_000, 200_000),
                     (hi + 5, 1000), (0, lo)):
        f = capi.LipFilter(T.LIP_BITVECTOR_EXACT, card, mn)
        assert f.build_from_table(table)
        o = oracle.LipFilter(T.LIP_BITVECTOR_EXACT, card, mn)
        inside = lambda k: k[(k >= mn) & (k < mn + card)]                     # (the reference's builder is handed keys of the range only)
        o.build(inside(keys), filter_bitmap=oracle.bitmap_from_bools(oracle.bools_from_bitmap(keep, keys.size)[(keys >= mn) & (keys < mn + card)]))
        o.build(inside(more))
        bm, cnt = f.probe(to_dev(probe, dev))
        ref = o.probe(probe)
        assert np.array_equal(bitmap_np(bm), ref), (mn, card)
        assert int(cnt.item()) == oracle.bitmap_count(ref, probe.size)
    f = capi.LipFilter(T.LIP_BITVECTOR_EXACT, hi - lo, lo)
    assert not f.build_from_table(table, num_new_keys=100)                    # 700 000 head words against 100 atomics: declined
    assert not capi.LipFilter(T.LIP_SINGLE_IDENTITY_HASH, 65_521, 0).build_from_table(table)
    hashed = capi.JoinTable(T.INT if dtype == np.int32 else T.LONG, keys.size)
    hashed.build(to_dev(keys, dev))
    assert not f.build_from_table(hashed)
    bm, cnt = f.probe(to_dev(probe, dev))
    assert int(cnt.item()) == 0                                               # the declined calls set nothing
    assert capi.lib.qsx_lip_build_from_join_table(None, table._h, -1, None) == T.ERR_INVALID_ARGUMENT
    assert capi.lib.qsx_lip_build_from_join_table(f._h, None, -1, None) == T.ERR_INVALID_ARGUMENT
    strided = capi.JoinTable(T.INT if dtype == np.int32 else T.LONG, 1000, key_range=(0, 7992), key_stride=8)   # one hash partition of a dense domain
    strided.build(to_dev((np.arange(1000) * 8).astype(dtype), dev))
    assert not f.build_from_table(strided)


@pytest.mark.parametrize("dtype", [np.int32, np.int64])
@pytest.mark.parametrize("kind", [T.LIP_BITVECTOR_EXACT, T.LIP_SINGLE_IDENTITY_HASH])
def test_lip_build_over_keys_in_order_merges_its_atomics_and_matches_oracle(capi, oracle, dev, dtype, kind):
    """A build side in key order (what dbgen writes): neighbouring lanes that set bits of one filter word hand them up the wave and
    one lane sets them (csrc/lip.hip lip_build_kernel).  Same filter words as the oracle for keys in order at every density —
    every key of the range (64 lanes, one word), one in ten, duplicates next to each other, runs that cross wave and group
    borders, holes punched by a filter bitmap, both ends of the declared range, and a word that repeats further down the wave
    (ascending runs laid side by side)."""
    rng = np.random.default_rng(11)
    card, mn = (300_000, 100) if kind == T.LIP_BITVECTOR_EXACT else (65_521, 0)
    cases = {
        "every key": np.arange(mn, mn + 250_000),
        "one in ten": np.sort(rng.choice(np.arange(mn, mn + 290_000), size=29_000, replace=False)),
        "duplicates": np.repeat(np.arange(mn + 5, mn + 40_005), 3),
        "both ends of the declared range": np.concatenate([np.arange(mn, mn + 700), np.arange(mn + card - 700, mn + card)]),
        "runs side by side": np.concatenate([np.arange(mn + o, mn + 200_000, 4) for o in (0, 1, 2, 3)]),
        "seventy keys": np.arange(mn + 130, mn + 200),
    }
    for name, k in cases.items():
        keys = k.astype(dtype)
        for filt in (None, oracle.bitmap_from_bools(rng.random(keys.size) < 0.6)):
            f = capi.LipFilter(kind, card, mn)
            o = oracle.LipFilter(kind, card, mn)
            f.build(to_dev(keys, dev), filter_bitmap=None if filt is None else bitmap_dev(filt, dev))
            o.build(keys, filter_bitmap=filt)
            probe = np.arange(mn - 50, mn + card + 50).astype(dtype)
            bm, cnt = f.probe(to_dev(probe, dev))
            ref = o.probe(probe)
            assert np.array_equal(bitmap_np(bm), ref), name
            assert int(cnt.item()) == oracle.bitmap_count(ref, probe.size), name


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


@pytest.mark.parametrize("P", [2, 5, 8])
@pytest.mark.parametrize("num_wide", [0, 7])
def test_partition_scatter_few_partitions_both_kernels(capi, oracle, dev, P, num_wide):
    """<= 8 partitions: the scatter kernel that stages all columns side by side (csrc/partition.hip,
    partition_scatter_small_kernel) while they fit 48 KiB of staging — key + 2-byte + 1-byte columns here — and the general
    kernel when they do not (seven more 8-byte columns).  Both stable, both equal to the oracle's partition; sizes on both
    sides of the chunking (one tile, a few thousand chunks, a ragged last tile)."""
    for n in (1, 2047, 2049, 300_001, 9_000_001):
        rng = np.random.default_rng(n + P + num_wide)
        keys = rng.integers(-50, 1 << 30, size=n).astype(np.int32)
        cols = [keys, rng.integers(0, 65535, size=n).astype(np.uint16), rng.integers(0, 255, size=n).astype(np.uint8)]
        cols += [rng.integers(0, 2**62, size=n).astype(np.int64) for _ in range(num_wide)]
        outs, offs = capi.partition_scatter(to_dev(keys, dev), P, [to_dev(c, dev) for c in cols])
        assert np.array_equal(offs.cpu().numpy(), oracle.partition_offsets(keys, P))
        for got, col in zip(outs, cols):
            assert np.array_equal(got.cpu().numpy(), oracle.partition_scatter(keys, P, col))


@pytest.mark.parametrize("dtype", [np.int32, np.int64])
@pytest.mark.parametrize("P,num_wide", [(1, 0), (2, 0), (3, 7), (8, 0), (8, 7), (41, 0), (64, 1)])
def test_partition_scatter_over_a_run_of_blocks(capi, oracle, dev, dtype, P, num_wide):
    """qsx_partition_scatter_blocks: the scatter reads every block's stripes where they lie and leaves what qsx_partition_scatter
    leaves for the rows laid end to end — equal to the oracle's stable partition of the concatenation.  Runs of equal blocks with
    a shorter last one (a stored relation), ragged runs with empty blocks, a lone block, blocks shorter than a tile and blocks
    longer than a workgroup's chunk (so that one block is many chunks and a chunk never straddles two blocks); all three
    scatter kernels (side-by-side staging for <= 8 partitions and narrow rows, the general kernel with and without ranks)."""
    shapes = [[349_525] * 6 + [12_345], [5, 0, 2047, 2048, 2049, 0, 70_001, 1], [1], [3_000_001, 17, 1_200_000], [0, 0], []]
    for rows in shapes:
        rng = np.random.default_rng(sum(rows) + P + num_wide)
        keys = [rng.integers(-1000, 1 << 30, size=r).astype(dtype) for r in rows]
        cols = [[k, rng.integers(0, 65535, size=k.size).astype(np.uint16), rng.integers(0, 255, size=k.size).astype(np.uint8)] +
                [rng.integers(0, 2**62, size=k.size).astype(np.int64) for _ in range(num_wide)] for k in keys]
        if not rows:
            continue   # (the Python mirror asks for one block at least; the C entry point with no blocks: test_gpu_edge_cases)
        outs, offs = capi.partition_scatter_blocks([to_dev(k, dev) for k in keys], P, [[to_dev(c, dev) for c in b] for b in cols])
        all_keys = np.concatenate(keys)
        assert np.array_equal(offs.cpu().numpy(), oracle.partition_offsets(all_keys, P))
        for c, got in enumerate(outs):
            col = np.concatenate([b[c] for b in cols])
            assert np.array_equal(got.cpu().numpy(), oracle.partition_scatter(all_keys, P, col) if all_keys.size else col)


@pytest.mark.parametrize("kind,anti,card", [(T.LIP_BITVECTOR_EXACT, False, 1_000_000), (T.LIP_BITVECTOR_EXACT, True, 999_937),
                                            (T.LIP_SINGLE_IDENTITY_HASH, False, 1_048_573), (T.LIP_BITVECTOR_EXACT, False, 3_000_000)])
def test_lip_probe_at_scale_lds_resident_filter(capi, oracle, dev, kind, anti, card):
    """>= 16 Mi probe rows against a filter of <= 1 Mi bits take the LDS-resident path (csrc/lip.hip); the
    3 M-bit filter stays in global memory.  Checked against the oracle on a 2 M-row slice and by the
    size-independent identity hits(all) = sum over slices."""
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    n = 20_000_003
    build = torch.randint(5, card - 5, (200_000,), device=dev, generator=g, dtype=torch.int32)
    probe = torch.randint(-1000, card + 1000, (n,), device=dev, generator=g, dtype=torch.int32)
    inb = torch.randint(-2**62, 2**62, ((n + 63) // 64,), device=dev, generator=g, dtype=torch.int64)
    inb[-1] = 0                                                     # trailing bits of the last word must be zero
    f = capi.LipFilter(kind, card, 0, anti)
    f.build(build)
    o = oracle.LipFilter(kind, card, 0, anti)
    o.build(build.cpu().numpy())
    for in_bitmap in (None, inb):
        bm, cnt = f.probe(probe, in_bitmap=in_bitmap)
        m = 2_000_000                                               # multiple of 64
        ref = o.probe(probe[:m].cpu().numpy(), in_bitmap=None if in_bitmap is None else bitmap_np(in_bitmap[: m // 64]))
        assert np.array_equal(bitmap_np(bm[: m // 64]), ref)
        assert int(cnt.item()) == int(capi.bitmap_count(bm, n).item())
        # the same rows through the small-input path (sliced below the LDS threshold) give the same words
        bm2, _ = f.probe(probe[:m], in_bitmap=None if in_bitmap is None else in_bitmap[: m // 64])
        assert torch.equal(bm2[: m // 64], bm[: m // 64])


@pytest.mark.parametrize("dtype", [np.int32, np.int64])
@pytest.mark.parametrize("kind,anti", [(T.LIP_SINGLE_IDENTITY_HASH, False), (T.LIP_BITVECTOR_EXACT, False),
                                       (T.LIP_BITVECTOR_EXACT, True)])
@pytest.mark.parametrize("rows", [[5000, 0, 1, 2047, 2048, 2049, 70_001, 0, 33], [8192] * 6 + [100], [20_000_000, 5]])
def test_lip_build_and_probe_over_runs_of_blocks(capi, oracle, dev, dtype, kind, anti, rows):
    """qsx_lip_build_blocks / qsx_lip_probe_blocks: one launch over a run of blocks (own key stripes and bitmaps, gaps in the
    bitmap lists) leaves the filter / gives the bitmaps of one call per block — also where the probe copies the filter to
    LDS (a run of >= 16 M rows)."""
    rng = np.random.default_rng(len(rows) + rows[0] % 97)
    build_blocks = [rng.integers(100, 5000, size=n).astype(dtype) for n in (3000, 0, 1, 513, 4096)]
    build_filters = [oracle.bitmap_from_bools(rng.random(b.size) < 0.9) if (i % 2 == 0 and b.size) else None for i, b in enumerate(build_blocks)]
    allb = np.concatenate(build_blocks)
    card, mn = (4099, 0) if kind == T.LIP_SINGLE_IDENTITY_HASH else (int(allb.max() - allb.min() + 1), int(allb.min()))
    f = capi.LipFilter(kind, card, mn, anti)
    o = oracle.LipFilter(kind, card, mn, anti)
    f.build_blocks([to_dev(b, dev) for b in build_blocks], filters=[None if m is None else bitmap_dev(m, dev) for m in build_filters])
    for b, m in zip(build_blocks, build_filters):
        o.build(b, filter_bitmap=m)
    probe_blocks = [rng.integers(-200, 6000, size=n).astype(dtype) for n in rows]
    in_bitmaps = [oracle.bitmap_from_bools(rng.random(n) < 0.6) if (i % 3 != 1 and n) else None for i, n in enumerate(rows)]
    dblocks = [to_dev(p, dev) for p in probe_blocks]
    for use_in in (False, True):
        outs, cnt = f.probe_blocks(dblocks, in_bitmaps=[None if m is None else bitmap_dev(m, dev) for m in in_bitmaps] if use_in else None)
        total = 0
        for i, p in enumerate(probe_blocks):
            ref = o.probe(p, in_bitmap=in_bitmaps[i] if use_in else None)
            if p.size:
                assert np.array_equal(bitmap_np(outs[i]), ref), (i, p.size, use_in)
            total += oracle.bitmap_count(ref, p.size)
        assert int(cnt.item()) == total
