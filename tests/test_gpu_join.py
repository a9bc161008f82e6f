"""GPU parity: K3/K4 (hash-join build + probe) through the C ABI against the
oracle's SimpleScalarSeparateChaining restatement.  Integer results are
bit-exact as sorted multisets of (probe_tid, build_tid): the reference leaves
pair order unspecified (relational_operators/tests/HashJoinOperator_unittest.cpp:480)."""
import numpy as np
import pytest
import torch

from quickstep_amd import types as T
from helpers import bitmap_dev, bitmap_np, sorted_pairs, to_dev

pytestmark = pytest.mark.gpu


FLAVOURS = ["hashed", "dense"]


def hip_join(capi, dev, key_type, build_blocks, probe_keys, est=None, capacity=None, build_filters=None,
             probe_filter=None, flavour="hashed"):
    total_build = sum(b.size for b in build_blocks)
    key_range = None
    if flavour == "dense":   # exact min/max statistics of the build side, as the optimizer would hold them
        allk = np.concatenate(build_blocks) if total_build else np.zeros(1, np.int64)
        key_range = (int(allk.min()), int(allk.max()))
    table = capi.JoinTable(key_type, total_build if est is None else est, key_range=key_range)
    base = 0
    for i, b in enumerate(build_blocks):
        f = None if build_filters is None else bitmap_dev(build_filters[i], dev)
        table.build(to_dev(b, dev), base_tid=base, filter_bitmap=f)
        base += b.size
    dp = to_dev(probe_keys, dev)
    pf = None if probe_filter is None else bitmap_dev(probe_filter, dev)
    total = int(table.probe_count(dp, filter_bitmap=pf).item())
    cap = total if capacity is None else capacity
    op, ob, cnt = table.probe(dp, capacity=cap, filter_bitmap=pf)
    assert int(cnt.item()) == total
    k = min(total, cap)
    return table, op.cpu().numpy()[:k], ob.cpu().numpy()[:k], total


def oracle_join(oracle, key_type, build_blocks, probe_keys, build_filters=None, probe_filter=None):
    t = oracle.JoinTable(key_type, sum(b.size for b in build_blocks))
    base = 0
    for i, b in enumerate(build_blocks):
        t.build(b, block_id=i, base_tid=base, filter_bitmap=None if build_filters is None else build_filters[i])
        base += b.size
    p, b = t.probe(probe_keys, filter_bitmap=probe_filter)
    return t, p, b


@pytest.mark.parametrize("flavour", FLAVOURS)
def test_golden_long_key_join(capi, dev, golden, flavour):
    g = golden["join_unittest"]
    bs = g["block_size"]
    dim = np.arange(g["num_dim_tuples"], dtype=np.int64)
    fact = np.arange(g["num_fact_tuples"], dtype=np.int64)
    blocks = [dim[b:b + bs] for b in range(0, dim.size, bs)]        # one build work order per 10-row block
    _, p, d, total = hip_join(capi, dev, T.LONG, blocks, fact, flavour=flavour)
    assert total == g["long_key"]["expected_num_results"]
    assert (np.bincount(d, minlength=dim.size) == g["long_key"]["expected_count_per_dim_long"]).all()
    assert np.array_equal(dim[d], fact[p])


@pytest.mark.parametrize("flavour", FLAVOURS)
def test_golden_int_duplicate_key_join(capi, dev, golden, flavour):
    g = golden["join_unittest"]
    bs = g["block_size"]
    dim_int = (np.arange(g["num_dim_tuples"]) % bs).astype(np.int32)
    fact_int = np.arange(g["num_fact_tuples"], dtype=np.int32)
    _, p, d, total = hip_join(capi, dev, T.INT, [dim_int[b:b + bs] for b in range(0, dim_int.size, bs)], fact_int,
                              flavour=flavour)
    e = g["int_duplicate_key"]
    assert total == e["expected_num_results"]
    assert (np.bincount(d, minlength=dim_int.size) == e["expected_count_per_dim_row"]).all()
    fc = np.bincount(p, minlength=fact_int.size)
    assert (fc[:bs] == e["expected_fact_count_first_rows"]).all() and (fc[bs:] == 0).all()


@pytest.mark.parametrize("flavour", FLAVOURS)
def test_cartesian_product_key(capi, oracle, dev, flavour):
    """CharKeyCartesianProductHashJoinTest shape (HashJoinOperator_unittest.cpp:692-826): one
    constant key on both sides, every dim row joins every fact row (200 x 300).  The
    constant stands in for the CHAR(\"100\") key; this drives the LDS-stage overflow path."""
    dim = np.full(200, 100, dtype=np.int32)
    fact = np.full(300, 100, dtype=np.int32)
    _, p, d, total = hip_join(capi, dev, T.INT, [dim], fact, flavour=flavour)
    assert total == 200 * 300
    assert (np.bincount(d, minlength=200) == 300).all() and (np.bincount(p, minlength=300) == 200).all()
    _, rp, rd = oracle_join(oracle, T.INT, [dim], fact)
    assert np.array_equal(sorted_pairs(p, d), sorted_pairs(rp, rd))


@pytest.mark.parametrize("key_type,dtype", [(T.INT, np.int32), (T.LONG, np.int64)])
@pytest.mark.parametrize("n_build,n_probe,key_range", [(1, 1, 1), (1000, 5000, 300), (50_000, 400_000, 100_000),
                                                         (200_000, 1_000_003, 150_000)])
@pytest.mark.parametrize("flavour", FLAVOURS)
def test_random_join_matches_oracle(capi, oracle, dev, key_type, dtype, n_build, n_probe, key_range, flavour):
    rng = np.random.default_rng(n_build + n_probe)
    lo = -key_range // 2                                  # negative keys: hash is the zero-extended pattern
    build = rng.integers(lo, lo + key_range, size=n_build).astype(dtype)
    probe = rng.integers(lo - 10, lo + key_range + 10, size=n_probe).astype(dtype)
    if dtype == np.int64 and flavour == "hashed":
        build[::7] += 2**40                               # keys beyond 32 bits
        probe[::5] += 2**40
    elif dtype == np.int64:
        build += 2**40                                    # dense: a bounded range far from zero
        probe += 2**40
        probe[::11] = -probe[::11]                        # far outside the range: no match
    blocks = np.array_split(build, 3)
    bf = [oracle.bitmap_from_bools(rng.random(b.size) < 0.8) if b.size else np.zeros(1, np.uint64) for b in blocks]
    pf = oracle.bitmap_from_bools(rng.random(n_probe) < 0.7)
    table, p, d, total = hip_join(capi, dev, key_type, blocks, probe, build_filters=bf, probe_filter=pf, flavour=flavour)
    otable, rp, rd = oracle_join(oracle, key_type, blocks, probe, build_filters=bf, probe_filter=pf)
    assert total == rp.size
    assert table.size() == otable.info()["buckets_allocated"]
    assert np.array_equal(sorted_pairs(p, d), sorted_pairs(rp, rd))
    # semi / anti existence bitmaps (HashJoinOperator.cpp:795-816, 860-877)
    for anti in (False, True):
        bm, cnt = table.probe_exists(to_dev(probe, dev), anti=anti, filter_bitmap=bitmap_dev(pf, dev))
        ref = otable.probe_exists(probe, anti=anti, filter_bitmap=pf)
        assert np.array_equal(bitmap_np(bm), ref)
        assert int(cnt.item()) == oracle.bitmap_count(ref, n_probe)


def test_empty_inputs(capi, dev):
    table = capi.JoinTable(T.INT, 0)
    empty = torch.empty(0, dtype=torch.int32, device=dev)
    table.build(empty)
    assert table.size() == 0
    keys = to_dev(np.arange(100, dtype=np.int32), dev)
    assert int(table.probe_count(keys).item()) == 0
    _, _, cnt = table.probe(keys)
    assert int(cnt.item()) == 0
    _, _, cnt = table.probe(empty)
    assert int(cnt.item()) == 0
    bm, c = table.probe_exists(keys, anti=True)
    assert int(c.item()) == 100


@pytest.mark.parametrize("flavour", FLAVOURS)
def test_table_grows_past_its_estimate(capi, oracle, dev, flavour):
    """est_entries = 4 but 300 k rows arrive in 6 build work orders: the table must
    resize like HashTable::resize (storage/HashTable.hpp:1437-1440) without losing entries."""
    rng = np.random.default_rng(9)
    build = rng.integers(0, 40_000, size=300_000).astype(np.int32)
    probe = rng.integers(0, 45_000, size=100_000).astype(np.int32)
    blocks = np.array_split(build, 6)
    table, p, d, total = hip_join(capi, dev, T.INT, blocks, probe, est=4, flavour=flavour)
    assert table.size() == build.size
    _, rp, rd = oracle_join(oracle, T.INT, blocks, probe)
    assert total == rp.size
    assert np.array_equal(sorted_pairs(p, d), sorted_pairs(rp, rd))


@pytest.mark.parametrize("flavour", FLAVOURS)
def test_capacity_smaller_than_matches_reports_full_count(capi, dev, flavour):
    build = np.arange(10_000, dtype=np.int32)
    probe = np.arange(10_000, dtype=np.int32)
    table, p, d, total = hip_join(capi, dev, T.INT, [build], probe, capacity=1234, flavour=flavour)
    assert total == 10_000 and p.size == 1234
    assert np.array_equal(build[d], probe[p])             # what was written is still valid pairs


def test_dense_table_rejects_build_keys_outside_its_range(capi, dev):
    """The dense flavour's precondition (exact min/max statistics, InjectJoinFilters.cpp:130-150): a build key
    outside the range is skipped and reported by the next synchronising call; probe keys outside simply miss."""
    table = capi.JoinTable(T.INT, 16, key_range=(10, 19))
    table.build(to_dev(np.array([10, 15, 19, 15], dtype=np.int32), dev))
    assert table.size() == 4
    p, b, cnt = table.probe(to_dev(np.array([9, 10, 15, 20, -5, 2**31 - 1], dtype=np.int32), dev))
    k = int(cnt.item())
    assert sorted(zip(p.cpu().numpy()[:k].tolist(), b.cpu().numpy()[:k].tolist())) == [(1, 0), (2, 1), (2, 3)]
    table.build(to_dev(np.array([20], dtype=np.int32), dev), base_tid=4)
    with pytest.raises(capi.QsxError):
        table.size()
    table.clear()
    assert table.size() == 0
    with pytest.raises(capi.QsxError):
        capi.JoinTable(T.LONG, 16, key_range=(0, 2**40))     # range too large for head words
    with pytest.raises(capi.QsxError):
        capi.JoinTable(T.INT, 16, key_range=(0, 100), key_stride=3)   # shift addressing: powers of two only


@pytest.mark.parametrize("key_type,dtype", [(T.INT, np.int32), (T.LONG, np.int64)])
def test_dense_table_over_one_hash_partition(capi, oracle, dev, key_type, dtype):
    """After the multi-GPU shuffle a rank holds the keys with key & (P-1) == rank of a dense domain
    (PartitionSchemeHeader.hpp:200-214): the strided dense table addresses them with (key - min) / P.
    Probe keys of other partitions (a mis-routed row would be one) must not match."""
    rng = np.random.default_rng(31)
    P, rank, domain = 8, 5, 400_000
    mine = np.arange(rank, domain, P)
    build = rng.choice(mine, size=60_000, replace=True).astype(dtype)      # duplicates included
    probe = rng.integers(-50, domain + 50, size=500_000).astype(dtype)     # all partitions + out of range
    table = capi.JoinTable(key_type, build.size, key_range=(int(mine[0]), int(mine[-1])), key_stride=P)
    table.build(to_dev(build, dev))
    assert table.size() == build.size
    dp = to_dev(probe, dev)
    total = int(table.probe_count(dp).item())
    p, b, cnt = table.probe(dp, capacity=total)
    _, rp, rd = oracle_join(oracle, key_type, [build], probe)
    assert total == rp.size == int(cnt.item())
    assert np.array_equal(sorted_pairs(p.cpu().numpy(), b.cpu().numpy()), sorted_pairs(rp, rd))
    table.build(to_dev(np.array([rank + 1], dtype=dtype), dev), base_tid=build.size)   # off the stride
    with pytest.raises(capi.QsxError):
        table.size()


def _table_geometry(est_entries, key_type):
    """quickstep_amd/csrc/join.hip capacity_for / home_bucket / fingerprint, restated for the test (INT and LONG keys)."""
    rows = max(512, est_entries)
    buckets = (rows * 5 + 63) // 64 + 1
    M64 = (1 << 64) - 1
    if key_type == T.INT:
        def home(keys):
            h = (keys.astype(np.uint64) & 0xFFFFFFFF) * 0x9E3779B9 & 0xFFFFFFFF
            return (h * buckets) >> 32
        def fingerprint(keys):
            f = ((keys.astype(np.uint64) & 0xFFFFFFFF) * 0x85EBCA6B & 0xFFFFFFFF) >> 24
            return np.where(f == 0, 1, f)
        return buckets, home, fingerprint
    def hashed(keys):                                   # mix64(k) * 0x9E3779B97F4A7C15 (Python integers: exact modulo 2^64)
        out = []
        for k in keys.tolist():
            k &= M64
            k ^= k >> 32
            k = k * 0x9E3779B97F4A7C15 & M64
            k ^= k >> 29
            out.append(k * 0x9E3779B97F4A7C15 & M64)
        return out
    def home(keys):
        return np.array([((h >> 32) * buckets) >> 32 for h in hashed(keys)], dtype=np.int64)
    def fingerprint(keys):
        return np.array([(h & 0xFF) or 1 for h in hashed(keys)], dtype=np.int64)
    return buckets, home, fingerprint


@pytest.mark.parametrize("key_type,dtype", [(T.INT, np.int32), (T.LONG, np.int64)])
@pytest.mark.parametrize("unique", [True, False])
def test_bucketed_table_crowded_buckets_and_shared_fingerprints(capi, oracle, dev, unique, key_type, dtype, monkeypatch):
    """The bucketed table against its worst inputs (QSX_JOIN_ADAPTIVE=0: no directly addressed shadow): hundreds of
    distinct keys whose home is ONE bucket (the sequence runs through a dozen full buckets, wrapping at the table's end),
    keys that share home bucket AND fingerprint (every fingerprint hit but one is a false positive), probes for keys that
    are absent but share bucket and fingerprint with present ones, and — unique=False — heavy duplicates of such keys."""
    monkeypatch.setenv("QSX_JOIN_ADAPTIVE", "0")
    est = 4_000
    buckets, home, fingerprint = _table_geometry(est, key_type)
    pool = np.arange(-3_000_000, 3_000_000, dtype=np.int64) if key_type == T.INT else np.arange(-600_000, 600_000, dtype=np.int64) * 1_000_003
    h, f = home(pool), fingerprint(pool)
    last = pool[h == buckets - 1][:300]                                  # one bucket, the LAST one: the walk wraps around
    twins = pool[(h == 7) & (f == 99)]                                   # same bucket, same fingerprint
    if twins.size < 40:                                                  # (the smaller LONG pool: any well-filled pair)
        pairs, counts = np.unique(np.stack([h, f], 1), axis=0, return_counts=True)
        hb, fb = pairs[np.argmax(counts)]
        twins = pool[(h == hb) & (f == fb)]
    assert last.size == 300 and twins.size >= 12
    rng = np.random.default_rng(5)
    spread = rng.choice(pool, size=2_500, replace=False)
    build = np.unique(np.concatenate([last[:200], twins[:twins.size // 2], spread])).astype(dtype)
    if not unique:
        build = np.concatenate([build, np.repeat(twins[:3].astype(dtype), 40), np.repeat(last[:2].astype(dtype), 25)])
    rng.shuffle(build)
    assert build.size <= est
    probe = np.concatenate([last, twins[:40], rng.choice(pool, size=50_000), build[:500]]).astype(dtype)   # present and absent look-alikes
    rng.shuffle(probe)
    table = capi.JoinTable(key_type, est)
    table.build(to_dev(build, dev))
    assert table.size() == build.size
    dp = to_dev(probe, dev)
    _, rp, rd = oracle_join(oracle, key_type, [build], probe)
    total = int(table.probe_count(dp).item())
    assert total == rp.size
    p, b, cnt = table.probe(dp, capacity=total)
    assert int(cnt.item()) == total
    assert np.array_equal(sorted_pairs(p.cpu().numpy()[:total], b.cpu().numpy()[:total]), sorted_pairs(rp, rd))
    exists, n_exist = table.probe_exists(dp)
    assert int(n_exist.item()) == np.isin(probe, build).sum()
    # the table grows (rehash into more buckets) and keeps every entry
    more = rng.choice(pool, size=30_000, replace=False).astype(dtype)
    table.build(to_dev(more, dev), base_tid=build.size)
    _, rp2, rd2 = oracle_join(oracle, key_type, [build, more], probe)
    total2 = int(table.probe_count(dp).item())
    assert total2 == rp2.size
    p, b, cnt = table.probe(dp, capacity=total2)
    assert np.array_equal(sorted_pairs(p.cpu().numpy()[:total2], b.cpu().numpy()[:total2]), sorted_pairs(rp2, rd2))
    table.close()


@pytest.mark.parametrize("flavour", FLAVOURS + ["hashed_no_shadow"])
def test_fk_join_at_scale_properties(capi, dev, flavour, monkeypatch):
    """C2 shape scaled to 1 M x 20 M (full 100 M runs in bench.py): every probe key hits
    exactly one build row, so pairs must be a permutation of the probe tids and satisfy the
    join condition; checked on device with size-independent reductions."""
    g = torch.Generator(device=dev)
    g.manual_seed(2)
    n_build, n_probe = 1_000_000, 20_000_000
    build = torch.randperm(n_build, device=dev, generator=g, dtype=torch.int32)
    probe = torch.randint(0, n_build, (n_probe,), device=dev, generator=g, dtype=torch.int32)
    if flavour == "hashed_no_shadow":
        monkeypatch.setenv("QSX_JOIN_ADAPTIVE", "0")      # the hashed kernels themselves (dense keys would get a shadow)
    table = capi.JoinTable(T.INT, n_build, key_range=(0, n_build - 1) if flavour == "dense" else None)
    table.build(build)
    p, b, cnt = table.probe(probe)
    assert int(cnt.item()) == n_probe
    assert bool((build[b.long()] == probe[p.long()]).all())
    assert int(p.long().sum().item()) == n_probe * (n_probe - 1) // 2
    assert int(torch.bincount(p.long(), minlength=n_probe).max().item()) == 1
    # 20 % match rate variant (keys drawn from 5x the build range)
    probe2 = torch.randint(0, 5 * n_build, (n_probe,), device=dev, generator=g, dtype=torch.int32)
    p2, b2, cnt2 = table.probe(probe2)
    k = int(cnt2.item())
    assert k == int((probe2 < n_build).sum().item())
    assert bool((build[b2[:k].long()] == probe2[p2[:k].long()]).all())


@pytest.mark.parametrize("key_type,dtype", [(T.INT, np.int32), (T.LONG, np.int64)])
def test_hashed_table_over_a_dense_key_domain_answers_from_its_shadow(capi, oracle, dev, key_type, dtype, monkeypatch):
    """A hashed table (no statistics from the optimizer) whose build keys turn out to span a small range copies its entries
    into a directly addressed shadow at the first probe (join.hip sealed_shadow).  Same pairs, counts and existence bitmaps
    with the shadow and without (QSX_JOIN_ADAPTIVE=0), with duplicates and filters, over runs of blocks, and again after
    further builds and after a clear (the shadow must not survive either)."""
    rng = np.random.default_rng(91)
    lo = -5_000 if dtype == np.int32 else 2**41
    n_build, n_probe = 120_000, 900_001
    build = (lo + rng.integers(0, 300_000, size=n_build)).astype(dtype)          # duplicates, range / entries = 2.5
    more = (lo + rng.integers(100_000, 420_000, size=70_000)).astype(dtype)      # a later build widens the range
    probe = (lo + rng.integers(-1000, 430_000, size=n_probe)).astype(dtype)
    pf = oracle.bitmap_from_bools(rng.random(n_probe) < 0.6)
    dp = to_dev(probe, dev)

    def check(table, build_blocks):
        for filt in (None, pf):
            fdev = None if filt is None else bitmap_dev(filt, dev)
            _, rp, rd = oracle_join(oracle, key_type, build_blocks, probe, probe_filter=filt)
            assert int(table.probe_count(dp, filter_bitmap=fdev).item()) == rp.size
            p, b, cnt = table.probe(dp, capacity=rp.size, filter_bitmap=fdev)
            assert int(cnt.item()) == rp.size
            assert np.array_equal(sorted_pairs(p.cpu().numpy()[:rp.size], b.cpu().numpy()[:rp.size]), sorted_pairs(rp, rd))
            bm, c = table.probe_exists(dp, filter_bitmap=fdev)
            want = np.zeros(n_probe, dtype=bool)
            want[rp] = True
            assert np.array_equal(bitmap_np(bm), oracle.bitmap_from_bools(want)) and int(c.item()) == int(want.sum())
        cuts = [0, 100_000, 100_000, 512_345, n_probe]
        blocks = [dp[a:b_] for a, b_ in zip(cuts[:-1], cuts[1:])]
        _, rp, rd = oracle_join(oracle, key_type, build_blocks, probe)
        p, b, cnt = table.probe_blocks(blocks, capacity=rp.size)
        assert int(cnt.item()) == rp.size
        assert np.array_equal(sorted_pairs(p.cpu().numpy()[:rp.size], b.cpu().numpy()[:rp.size]), sorted_pairs(rp, rd))

    for adaptive in ("1", "0"):
        monkeypatch.setenv("QSX_JOIN_ADAPTIVE", adaptive)
        table = capi.JoinTable(key_type, n_build)
        table.build(to_dev(build, dev))
        check(table, [build])
        table.build(to_dev(more, dev), base_tid=n_build)          # after a probe: the next probe looks at the bounds again
        check(table, [build, more])
        table.clear()
        table.build(to_dev(more, dev))
        check(table, [more])
        table.close()


@pytest.mark.parametrize("key_type,dtype", [(T.INT, np.int32), (T.LONG, np.int64)])
def test_dense_table_packs_its_head_words_for_the_probes(capi, oracle, dev, key_type, dtype, monkeypatch):
    """A directly addressed table of about the size of an XCD's L2 (1 M key values: 4 MiB of head words) is probed through a
    3-byte copy of its head words (join.hip sealed_pack).  Same pairs / counts / bitmaps with the copy and without
    (QSX_JOIN_ADAPTIVE=0): duplicate keys (chains), a probe filter, runs of blocks, a further build after a probe, a clear —
    and tuple ids beyond 2^23 (a large base_tid) must keep the table on its 4-byte words."""
    rng = np.random.default_rng(17)
    lo = 7 if dtype == np.int32 else 2**40
    span, n_build, n_probe = 1_000_000, 300_000, 1_200_003
    build = (lo + rng.integers(0, span, size=n_build)).astype(dtype)           # ~4 % of the keys twice or more
    more = (lo + rng.integers(0, span, size=50_000)).astype(dtype)
    probe = (lo + rng.integers(-100, span + 100, size=n_probe)).astype(dtype)
    pf = oracle.bitmap_from_bools(rng.random(n_probe) < 0.5)
    dp = to_dev(probe, dev)

    def check(table, blocks, bases):
        t = oracle.JoinTable(key_type, sum(b.size for b in blocks))
        for b, base in zip(blocks, bases):
            t.build(b, base_tid=base)
        for filt in (None, pf):
            fdev = None if filt is None else bitmap_dev(filt, dev)
            rp, rd = t.probe(probe, filter_bitmap=filt)
            assert int(table.probe_count(dp, filter_bitmap=fdev).item()) == rp.size
            p, b, cnt = table.probe(dp, capacity=rp.size, filter_bitmap=fdev)
            assert int(cnt.item()) == rp.size
            assert np.array_equal(sorted_pairs(p.cpu().numpy()[:rp.size], b.cpu().numpy()[:rp.size]), sorted_pairs(rp, rd))
        bm, c = table.probe_exists(dp)
        want = np.zeros(n_probe, dtype=bool)
        want[t.probe(probe)[0]] = True
        assert np.array_equal(bitmap_np(bm), oracle.bitmap_from_bools(want)) and int(c.item()) == int(want.sum())
        rp, rd = t.probe(probe)
        p, b, cnt = table.probe_blocks([dp[:400_001], dp[400_001:]], capacity=rp.size)
        assert np.array_equal(sorted_pairs(p.cpu().numpy()[:rp.size], b.cpu().numpy()[:rp.size]), sorted_pairs(rp, rd))

    for adaptive in ("1", "0"):
        monkeypatch.setenv("QSX_JOIN_ADAPTIVE", adaptive)
        table = capi.JoinTable(key_type, n_build, key_range=(lo, lo + span - 1))
        table.build(to_dev(build, dev))
        check(table, [build], [0])
        table.build(to_dev(more, dev), base_tid=n_build)
        check(table, [build, more], [0, n_build])
        table.clear()
        table.build(to_dev(more, dev), base_tid=(1 << 23) + 5)          # tuple ids that do not fit 23 bits
        check(table, [more], [(1 << 23) + 5])
        table.close()


def hip_composite_join(capi, dev, build_cols, probe_cols):
    """Composite-key inner join through the C ABI: fold the components into one LONG key
    (qsx_join_key_pack), single-key table, and — when the fold is a hash, not an exact packing —
    verify the components of every returned pair (gather + attribute-vs-attribute K1 + K2)."""
    bd = [to_dev(c, dev) for c in build_cols]
    pd = [to_dev(c, dev) for c in probe_cols]
    bk, exact = capi.join_key_pack(bd)
    pk, exact2 = capi.join_key_pack(pd)
    assert exact == exact2
    table = capi.JoinTable(T.LONG, bk.numel())
    table.build(bk)
    total = int(table.probe_count(pk).item())
    p, b, cnt = table.probe(pk, capacity=total)
    assert int(cnt.item()) == total
    p, b = p[:total], b[:total]
    if not exact and total > 0:
        bm = None
        for pc, bc in zip(pd, bd):
            bm, _ = capi.select_cmp_columns(capi.gather(pc, p), capi.gather(bc, b), T.EQ, filter_bitmap=bm)
        (p, b), k = capi.compact_gather([p, b], bm, total)
        p, b = p[:int(k.item())], b[:int(k.item())]
    return p, b, exact


@pytest.mark.parametrize("second_type", [np.int32, np.int64])
def test_golden_composite_key_join_and_residual(capi, dev, golden, second_type):
    """CompositeKeyHashJoinTest (:999-1177) and ...WithResidualPredicateTest (:1187-1375): keys
    (long, varchar) where the VARCHAR holds tid/2*2 (dim) and tid (fact) as digits; the second
    component is carried as an integer here.  (LONG, INT) needs 12 bytes -> hashed fold + component
    verification; with both components narrowed to INT the fold is an exact packing — both run."""
    g = golden["join_unittest"]
    dim = np.arange(g["num_dim_tuples"])
    fact = np.arange(g["num_fact_tuples"])
    first = np.int64 if second_type == np.int64 else np.int32
    build_cols = [dim.astype(first), (dim // 2 * 2).astype(second_type)]
    probe_cols = [fact.astype(first), fact.astype(second_type)]
    p, b, exact = hip_composite_join(capi, dev, build_cols, probe_cols)
    assert exact == (second_type == np.int32)
    p, b = p.cpu().numpy(), b.cpu().numpy()
    assert p.size == g["composite_key"]["expected_num_results"]
    assert np.array_equal(np.sort(b), np.arange(0, dim.size, 2)) and np.array_equal(np.sort(p), np.arange(0, dim.size, 2))
    # residual predicate dim.long < 15 evaluated on the joined pairs (HashJoinOperator.cpp:510-524)
    pt, bt = to_dev(p, dev), to_dev(b, dev)
    dim_long = capi.gather(to_dev(dim.astype(np.int64), dev), bt)
    bm, cnt = capi.select_cmp(dim_long, T.LT, g["composite_key_residual"]["residual_dim_long_less_than"])
    assert int(cnt.item()) == g["composite_key_residual"]["expected_num_results"]
    (fp, fb), k = capi.compact_gather([pt, bt], bm, p.size)
    k = int(k.item())
    assert sorted(fb[:k].cpu().tolist()) == list(range(0, 15, 2)) == sorted(fp[:k].cpu().tolist())


@pytest.mark.parametrize("types", [(np.int32, np.int32), (np.int64, np.int32), (np.int64, np.int64, np.int32)])
def test_random_composite_join_matches_oracle(capi, oracle, dev, types):
    rng = np.random.default_rng(len(types) * 17 + 3)
    n_build, n_probe = 30_000, 200_000
    # small component domains: plenty of duplicate composite keys and, for the hashed fold,
    # rows that agree in one component only
    build_cols = [rng.integers(-20, 20, size=n_build).astype(t) for t in types]
    probe_cols = [rng.integers(-22, 22, size=n_probe).astype(t) for t in types]
    p, b, exact = hip_composite_join(capi, dev, build_cols, probe_cols)
    assert exact == (sum(np.dtype(t).itemsize for t in types) <= 8)
    ot = oracle.CompositeJoinTable([T.INT if t == np.int32 else T.LONG for t in types], n_build)
    ot.build(build_cols)
    rp, rb = ot.probe(probe_cols)
    assert p.numel() == rp.size
    assert np.array_equal(sorted_pairs(p.cpu().numpy(), b.cpu().numpy()), sorted_pairs(rp, rb))
    if not exact:
        # the device fold is the reference's composite hash (HashTable.hpp:2109-2119)
        keys, _ = capi.join_key_pack([to_dev(c[:64], dev) for c in build_cols])
        assert np.array_equal(keys.cpu().numpy().view(np.uint64), ot.hash_rows([c[:64] for c in build_cols]))


def test_semi_join_with_residual_predicate(capi, oracle, dev):
    """HashSemiJoinWorkOrder::executeWithResidualPredicate (HashJoinOperator.cpp:680-793): a probe row
    qualifies when at least one of its pairs passes the residual (here probe.v < build.w)."""
    rng = np.random.default_rng(5)
    n_build, n_probe = 20_000, 150_000
    bk = rng.integers(0, 5000, size=n_build).astype(np.int32)
    bw = rng.integers(0, 100, size=n_build).astype(np.int64)
    pk = rng.integers(0, 6000, size=n_probe).astype(np.int32)
    pv = rng.integers(0, 100, size=n_probe).astype(np.int64)
    table, p, b, total = hip_join(capi, dev, T.INT, [bk], pk)
    pt, bt = to_dev(p, dev), to_dev(b, dev)
    bm, _ = capi.select_cmp_columns(capi.gather(to_dev(pv, dev), pt), capi.gather(to_dev(bw, dev), bt), T.LT)
    (kept, ), k = capi.compact_gather([pt], bm, total)
    got = bitmap_np(capi.tids_to_bitmap(kept[:int(k.item())], n_probe))
    _, rp, rb = oracle_join(oracle, T.INT, [bk], pk)
    keep = pv[rp] < bw[rb]
    want = oracle.tids_to_bitmap(rp[keep], n_probe)
    assert np.array_equal(got, want)
    assert np.array_equal(want, oracle.bitmap_from_bools(np.isin(np.arange(n_probe), rp[keep])))
    # the anti join is the complement within the probe block (:860-877 with residual :930-1000)
    anti = bitmap_np(capi.bitmap_combine(3, capi.tids_to_bitmap(kept[:int(k.item())], n_probe), None, n_probe))
    assert oracle.bitmap_count(anti, n_probe) == n_probe - oracle.bitmap_count(want, n_probe)


PROBE_RUNS = {
    "ragged": [5000, 0, 1, 4095, 4096, 4097, 70_001, 0, 33],
    "equal": [12_288] * 7,
    "equal_short_last": [8192] * 5 + [77],
    "one": [300_001],
}


@pytest.mark.parametrize("table_kind", ["dense", "hashed_dense_keys", "hashed_sparse_keys"])
@pytest.mark.parametrize("key_type,dtype", [(T.INT, np.int32), (T.LONG, np.int64)])
@pytest.mark.parametrize("duplicates,cover", [(False, "1"), (False, "0"), (True, "1")])
def test_probe_writes_the_projected_output_relation(capi, oracle, dev, table_kind, key_type, dtype, duplicates, cover, monkeypatch):
    """qsx_join_probe_project_blocks: the tuples HashInnerJoinWorkOrder materialises from its pair list
    (HashJoinOperator.cpp:494-560), written by the probe itself — checked as a multiset of rows against the oracle's pairs
    joined with the same attributes on the host.  Directly addressed tables and shadows run the fused kernel, a hashed
    table over spread keys the pair list + gathers inside the call.  Build side in several segments (uniform and ragged),
    probe side in ragged blocks with an empty one, attribute widths 1 / 2 / 4 / 8, a filter, a capacity below the result.
    Unique build keys get a covering array of the build-side attributes (entries of 16 bytes for the INT key's projection, 4 or
    8 for the second one; the LONG key's first projection is too wide for one); QSX_JOIN_COVER=0 keeps head[] + stripes."""
    monkeypatch.setenv("QSX_JOIN_COVER", cover)
    rng = np.random.default_rng(97 + duplicates)
    nbuild = 70_000 if table_kind != "dense" or not duplicates else 9_000
    spread = table_kind == "hashed_sparse_keys"
    bkeys = rng.permutation(nbuild).astype(np.int64)
    if duplicates:
        bkeys[rng.integers(0, nbuild, size=nbuild // 10)] = bkeys[rng.integers(0, nbuild, size=nbuild // 10)]
    if spread:
        bkeys = bkeys * 30_011 - 10**9
    bkeys = bkeys.astype(dtype)
    for seg_rows in ([nbuild], [nbuild // 4] * 3 + [nbuild - 3 * (nbuild // 4)], [1000, 0, nbuild - 5000, 4000]):
        first_tids = [100 + int(x) for x in np.cumsum([0] + seg_rows[:-1])]     # (tuple ids need not start at 0)
        cuts = np.cumsum([0] + seg_rows)
        b_long = rng.integers(-2**60, 2**60, size=nbuild)
        b_short = rng.integers(0, 60000, size=nbuild).astype(np.uint16)
        key_range = (int(bkeys.min()), int(bkeys.max())) if table_kind == "dense" else None
        table = capi.JoinTable(key_type, nbuild, key_range=key_range)
        dsegs = [to_dev(bkeys[a:b], dev) for a, b in zip(cuts[:-1], cuts[1:])]
        for sg, keys in enumerate(dsegs):
            if keys.numel():
                table.build(keys, base_tid=first_tids[sg])
        rows = [5000, 0, 123, 40_000, 4097]
        pkeys = [(rng.integers(-5, nbuild + nbuild // 3, size=n) * (30_011 if spread else 1) - (10**9 if spread else 0)).astype(dtype)
                 for n in rows]
        p_double = [rng.normal(size=n) for n in rows]
        p_byte = [rng.integers(0, 250, size=n).astype(np.uint8) for n in rows]
        p_int = [rng.integers(-2**31, 2**31 - 1, size=n).astype(np.int32) for n in rows]
        keep = [rng.random(n) < 0.8 for n in rows]
        for use_filter in (False, True):
            filters = [bitmap_dev(oracle.bitmap_from_bools(m), dev) if m.size else None for m in keep] if use_filter else None
            # expected: pairs on the host
            allp = np.concatenate(pkeys)
            order = np.argsort(bkeys, kind="stable")
            lo, hi = np.searchsorted(bkeys[order], allp, "left"), np.searchsorted(bkeys[order], allp, "right")
            live = np.concatenate(keep) if use_filter else np.ones(allp.size, dtype=bool)
            counts = np.where(live, hi - lo, 0)
            probe_rows = np.repeat(np.arange(allp.size), counts)
            build_rows = order[np.concatenate([np.arange(a, b) for a, b, c in zip(lo, hi, counts) if c > 0])] if counts.sum() else np.zeros(0, dtype=np.int64)
            want = np.stack([np.concatenate(p_double)[probe_rows].view(np.int64), np.concatenate(p_byte)[probe_rows].astype(np.int64),
                             np.concatenate(p_int)[probe_rows].astype(np.int64), allp[probe_rows].astype(np.int64),
                             b_long[build_rows], b_short[build_rows].astype(np.int64), bkeys[build_rows].astype(np.int64)], axis=1)
            outs, cnt = table.probe_project_blocks(
                [to_dev(k, dev) for k in pkeys],
                [[to_dev(x, dev) for x in col] for col in (p_double, p_byte, p_int, pkeys)],
                [[to_dev(col[a:b], dev) for a, b in zip(cuts[:-1], cuts[1:])] for col in (b_long, b_short, bkeys)],
                build_first_tids=first_tids, filters=filters)
            k = int(cnt.item())
            assert k == want.shape[0]
            got = np.stack([outs[0][:k].cpu().numpy().view(np.int64)] + [o[:k].cpu().numpy().astype(np.int64) for o in outs[1:]], axis=1)
            assert np.array_equal(got[np.lexsort(got.T[::-1])], want[np.lexsort(want.T[::-1])])
        # a capacity below the result: the count is the full one, what fits is a subset of the result's rows
        small = max(1, want.shape[0] // 3)
        outs, cnt = table.probe_project_blocks(
            [to_dev(k, dev) for k in pkeys], [[to_dev(x, dev) for x in col] for col in (p_int, pkeys)],
            [[to_dev(col[a:b], dev) for a, b in zip(cuts[:-1], cuts[1:])] for col in (bkeys,)],
            build_first_tids=first_tids, filters=filters, capacity=small)
        assert int(cnt.item()) == want.shape[0]
        got = np.stack([o[:small].cpu().numpy().astype(np.int64) for o in outs], axis=1)
        assert np.array_equal(got[:, 1], got[:, 2])
        table.close()
    # nothing to probe
    table = capi.JoinTable(key_type, 10)
    none = to_dev(np.zeros(0, dtype=dtype), dev)
    outs, cnt = table.probe_project_blocks([none], [[none]], [[none]])
    assert int(cnt.item()) == 0


@pytest.mark.parametrize("shape", sorted(PROBE_RUNS))
@pytest.mark.parametrize("key_type,dtype", [(T.INT, np.int32), (T.LONG, np.int64)])
@pytest.mark.parametrize("flavour", FLAVOURS)
def test_probe_over_a_run_of_blocks_equals_block_by_block(capi, oracle, dev, flavour, key_type, dtype, shape, monkeypatch):
    """qsx_join_probe_blocks / qsx_join_probe_exists_blocks: one launch over a run of probe blocks gives the pairs and
    bitmaps of one probe per block (duplicate build keys, per-block filters with gaps, explicit and run-global tids)."""
    rows = PROBE_RUNS[shape]
    rng = np.random.default_rng(len(rows) * 7 + rows[0])
    build = rng.integers(0, 3000, size=9000).astype(dtype)          # ~3 entries per key: chains / multi-unit buckets
    blocks = [rng.integers(-5, 3300, size=n).astype(dtype) for n in rows]
    filters = [oracle.bitmap_from_bools(rng.random(n) < 0.5) if (i % 3 != 2 and n) else None for i, n in enumerate(rows)]
    key_range = (0, 2999) if flavour == "dense" else None
    table = capi.JoinTable(key_type, build.size, key_range=key_range)
    table.build(to_dev(build, dev))
    ot = oracle.JoinTable(key_type, build.size)
    ot.build(build, block_id=0, base_tid=0)
    dblocks = [to_dev(b, dev) for b in blocks]
    dfilters = [None if f is None else bitmap_dev(f, dev) for f in filters]
    bases = [int(x) for x in np.cumsum([0] + rows[:-1]) * 2 + 11]      # explicit tids: gaps between the blocks
    for use_filters in (False, True):
        for two_pass in ("0", "1"):
            monkeypatch.setenv("QSX_JOIN_TWO_PASS", two_pass)
            for base_tids in (None, bases):
                want_p, want_b = [], []
                start = 0
                for i, blk in enumerate(blocks):
                    rp, rb = ot.probe(blk, filter_bitmap=filters[i] if use_filters else None)
                    want_p.append(rp + (start if base_tids is None else base_tids[i]))
                    want_b.append(rb)
                    start += blk.size
                want_p, want_b = np.concatenate(want_p), np.concatenate(want_b)
                p, b, cnt = table.probe_blocks(dblocks, capacity=want_p.size + 5, base_tids=base_tids,
                                               filters=dfilters if use_filters else None)
                k = int(cnt.item())
                assert k == want_p.size, (shape, use_filters, two_pass)
                assert int(table.probe_count_blocks(dblocks, filters=dfilters if use_filters else None).item()) == k
                assert np.array_equal(sorted_pairs(p.cpu().numpy()[:k], b.cpu().numpy()[:k]), sorted_pairs(want_p, want_b))
        monkeypatch.delenv("QSX_JOIN_TWO_PASS")
        for anti in (False, True):
            outs, cnt = table.probe_exists_blocks(dblocks, anti=anti, filters=dfilters if use_filters else None)
            total = 0
            for i, blk in enumerate(blocks):
                ref = ot.probe_exists(blk, anti=anti, filter_bitmap=filters[i] if use_filters else None)
                if blk.size:
                    assert np.array_equal(bitmap_np(outs[i]), ref), (shape, i, anti, use_filters)
                total += oracle.bitmap_count(ref, blk.size)
            assert int(cnt.item()) == total
    # a pair list that is too small still reports the full count
    p, b, cnt = table.probe_blocks(dblocks, capacity=3)
    assert int(cnt.item()) == sum(ot.probe(blk)[0].size for blk in blocks)
    p, b, cnt = table.probe_blocks([])
    assert int(cnt.item()) == 0


@pytest.mark.parametrize("num_blocks", [3, 64, 65, 700])
def test_run_global_tids_feed_the_segmented_gathers(capi, oracle, dev, num_blocks):
    """The pair list of qsx_join_probe_blocks numbers probe rows through the run; qsx_gather_segmented /
    qsx_bitmap_gather_segmented (device-resident segment table beyond 64 segments) fetch the probe side's values and null
    bits from the blocks' own stripes."""
    rng = np.random.default_rng(num_blocks)
    rows = [int(x) for x in rng.integers(0, 900, size=num_blocks)]
    rows[0] = 5000
    build = rng.permutation(4000).astype(np.int32)
    table = capi.JoinTable(T.INT, build.size, key_range=(0, 3999))
    table.build(to_dev(build, dev))
    keys = [rng.integers(0, 8000, size=n).astype(np.int32) for n in rows]
    payload = [rng.normal(size=n) for n in rows]
    narrow = [rng.integers(0, 200, size=n).astype(np.uint8) for n in rows]
    nulls = [rng.random(n) < 0.2 if b % 3 else None for b, n in enumerate(rows)]
    p, b, cnt = table.probe_blocks([to_dev(k, dev) for k in keys])
    k = int(cnt.item())
    allk = np.concatenate(keys)
    assert k == int((allk < 4000).sum())
    first = [int(x) for x in np.cumsum([0] + rows[:-1])]
    pt = p[:k].contiguous()
    got_pay = capi.gather_segmented([to_dev(x, dev) for x in payload], first, pt).cpu().numpy()
    got_narrow = capi.gather_segmented([to_dev(x, dev) for x in narrow], first, pt).cpu().numpy()
    got_key = capi.gather_segmented([to_dev(x, dev) for x in keys], first, pt).cpu().numpy()
    hp, hb = pt.cpu().numpy(), b[:k].cpu().numpy()
    assert np.array_equal(got_key, allk[hp]) and np.array_equal(build[hb], allk[hp])
    assert np.array_equal(got_pay, np.concatenate(payload)[hp]) and np.array_equal(got_narrow, np.concatenate(narrow)[hp])
    segs = [None if m is None or m.size == 0 else bitmap_dev(oracle.bitmap_from_bools(m), dev) for m in nulls]
    whole = np.concatenate([np.zeros(n, dtype=bool) if m is None else m for m, n in zip(nulls, rows)])
    with_padding = torch.cat([pt, torch.full((3,), -1, dtype=torch.int32, device=pt.device)])
    out = capi.bitmap_gather_segmented(segs, first, with_padding)
    want = oracle.bitmap_from_bools(np.concatenate([whole[hp], np.ones(3, dtype=bool)]))
    assert np.array_equal(out.cpu().numpy().view(np.uint64), want[:(k + 3 + 63) // 64])


@pytest.mark.parametrize("shape", sorted(PROBE_RUNS))
@pytest.mark.parametrize("key_type,dtype", [(T.INT, np.int32), (T.LONG, np.int64)])
@pytest.mark.parametrize("flavour", FLAVOURS)
def test_build_over_a_run_of_blocks_equals_block_by_block(capi, oracle, dev, flavour, key_type, dtype, shape):
    """qsx_join_build_blocks: one launch over a run of build blocks (own stripes, filters with gaps, relation-global base
    tids) leaves the table of one build per block — same entry count, same pairs for any probe; the hashed table also
    grows past its estimate inside the call."""
    rows = PROBE_RUNS[shape]
    rng = np.random.default_rng(len(rows) * 3 + rows[0])
    blocks = [rng.integers(0, 5000, size=n).astype(dtype) for n in rows]      # duplicates within and across blocks
    filters = [oracle.bitmap_from_bools(rng.random(n) < 0.7) if (i % 2 == 0 and n) else None for i, n in enumerate(rows)]
    bases = [int(x) for x in np.cumsum([0] + rows[:-1])]
    probe = rng.integers(-3, 5200, size=40_000).astype(dtype)
    for use_filters in (False, True):
        table = capi.JoinTable(key_type, 64 if flavour == "hashed" else sum(rows), key_range=(0, 4999) if flavour == "dense" else None)
        table.build_blocks([to_dev(b, dev) for b in blocks], bases,
                           filters=[None if f is None else bitmap_dev(f, dev) for f in filters] if use_filters else None)
        ot = oracle.JoinTable(key_type, sum(rows))
        for i, b in enumerate(blocks):
            ot.build(b, block_id=i, base_tid=bases[i], filter_bitmap=filters[i] if use_filters else None)
        assert table.size() == ot.info()["buckets_allocated"]
        rp, rb = ot.probe(probe)
        p, b, cnt = table.probe(to_dev(probe, dev), capacity=rp.size + 1)
        k = int(cnt.item())
        assert k == rp.size
        assert np.array_equal(sorted_pairs(p.cpu().numpy()[:k], b.cpu().numpy()[:k]), sorted_pairs(rp, rb))


@pytest.mark.parametrize("types", [(np.int32, np.int32), (np.int64, np.int32), (np.int64, np.int64), (np.int32, np.int32, np.int32)])
def test_composite_keys_of_a_run_of_blocks_pack_like_block_by_block(capi, dev, types):
    """qsx_join_key_pack_blocks: the packed keys of a run (one stripe, block after block) = qsx_join_key_pack per block
    (exact 8-byte packing and the reference's CombineHashes fold for wider keys)."""
    rng = np.random.default_rng(len(types) * 10 + types[0]().itemsize)
    rows = [3000, 0, 1, 511, 512, 513, 70_001, 5]
    blocks = [[to_dev(rng.integers(-1000, 1000, size=n).astype(t), dev) for t in types] for n in rows]
    packed, exact = capi.join_key_pack_blocks(blocks)
    want = torch.cat([capi.join_key_pack(b)[0] for b in blocks if b[0].numel()])
    assert exact == capi.join_key_pack([b for b in blocks if b[0].numel()][0])[1]
    assert torch.equal(packed, want)


@pytest.mark.parametrize("key_type,dtype", [(T.INT, np.int32), (T.LONG, np.int64)])
@pytest.mark.parametrize("key_range", [1, 7, 2_000, 11_000, 20_000, 36 * 1024, 36 * 1024 + 1])
@pytest.mark.parametrize("flavour", ["dense", "hashed"])
def test_small_build_sides_are_probed_from_lds(capi, oracle, dev, key_type, dtype, key_range, flavour, monkeypatch):
    """csrc/join_lds.hpp: a directly addressed table of up to 36 Ki key values (or the shadow of a hashed table over such a
    domain) is copied into LDS by every workgroup of a long probe.  Key ranges that make one, two and four workgroups per
    CU (1024 / 512 / 256 threads), the largest range the LDS takes and the first one it does not; duplicate build keys
    (chains through the overflow list), probe keys outside the range, a probe filter (NULL probe keys reach the kernels as
    one), count / pairs / existence / anti, a run of blocks — all equal to the oracle, and to the same calls with
    QSX_JOIN_LDS=0."""
    rng = np.random.default_rng(1000 + key_range)
    lo = -77 if dtype == np.int32 else 2**40
    n_probe = max(300_000, 40 * key_range)
    n_build = max(1, int(key_range * 1.2)) if key_range < 30_000 else key_range // 2
    build = (lo + rng.integers(0, key_range, size=n_build)).astype(dtype)     # duplicates (1.2 rows per key value) or holes
    build[0], build[-1] = lo, lo + key_range - 1                              # the whole range is in use
    probe = (lo + rng.integers(-key_range // 10 - 3, key_range + key_range // 10 + 3, size=n_probe)).astype(dtype)
    pf = oracle.bitmap_from_bools(rng.random(n_probe) < 0.7)
    dp = to_dev(probe, dev)
    results = {}
    for lds in ("1", "0"):
        monkeypatch.setenv("QSX_JOIN_LDS", lds)
        table = capi.JoinTable(key_type, n_build, key_range=(lo, lo + key_range - 1) if flavour == "dense" else None)
        table.build(to_dev(build, dev))
        got = []
        for filt in (None, pf):
            fdev = None if filt is None else bitmap_dev(filt, dev)
            _, rp, rd = oracle_join(oracle, key_type, [build], probe, probe_filter=filt)
            assert int(table.probe_count(dp, filter_bitmap=fdev).item()) == rp.size
            p, b, cnt = table.probe(dp, capacity=rp.size, filter_bitmap=fdev)
            assert int(cnt.item()) == rp.size
            pairs = sorted_pairs(p.cpu().numpy()[:rp.size], b.cpu().numpy()[:rp.size])
            assert np.array_equal(pairs, sorted_pairs(rp, rd))
            want = np.zeros(n_probe, dtype=bool)
            want[rp] = True
            live = np.ones(n_probe, dtype=bool) if filt is None else oracle.bools_from_bitmap(filt, n_probe)
            for anti in (False, True):
                bm, c = table.probe_exists(dp, anti=anti, filter_bitmap=fdev)
                ref = (want != anti) & live
                assert np.array_equal(bitmap_np(bm), oracle.bitmap_from_bools(ref)) and int(c.item()) == int(ref.sum())
            got.append(pairs)
        cuts = [0, 4096, 4096, 123_457, n_probe]
        blocks = [dp[a:b_] for a, b_ in zip(cuts[:-1], cuts[1:])]
        _, rp, rd = oracle_join(oracle, key_type, [build], probe)
        p, b, cnt = table.probe_blocks(blocks, capacity=rp.size)
        assert int(cnt.item()) == rp.size
        assert np.array_equal(sorted_pairs(p.cpu().numpy()[:rp.size], b.cpu().numpy()[:rp.size]), sorted_pairs(rp, rd))
        results[lds] = got
        table.close()
    for a, b in zip(results["1"], results["0"]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("key_type,dtype", [(T.INT, np.int32), (T.LONG, np.int64)])
@pytest.mark.parametrize("n_keys,duplicates", [(1, False), (40, True), (3_000, False), (3_000, True), (6_000, False), (12_500, False), (14_000, False)])
def test_small_bucketed_tables_are_probed_from_lds(capi, oracle, dev, key_type, dtype, n_keys, duplicates, monkeypatch):
    """csrc/join_lds_bucket.hpp: a bucketed table (sparse keys: no shadow) whose slots and fingerprint plane fit 144 KiB is
    copied into LDS by every workgroup of a long probe — INT tables up to ~13 K keys, LONG tables up to ~6.5 K; the sizes
    beyond stay in L2.  Unique and duplicate build keys (a duplicate-free INT table ends a probe at its first match), keys
    that are absent, a probe filter, count / pairs / existence / anti and a run of blocks: equal to the oracle and to the
    same calls with QSX_JOIN_LDS=0."""
    rng = np.random.default_rng(7000 + n_keys + (1 if duplicates else 0))
    info = np.iinfo(dtype)
    distinct = rng.choice(np.arange(-2**30, 2**30, dtype=np.int64) if dtype == np.int32 else rng.integers(info.min // 2, info.max // 2, size=4 * n_keys + 8),
                          size=max(2, 2 * n_keys), replace=False).astype(dtype)
    present, absent = distinct[:n_keys], distinct[n_keys:]
    build = np.concatenate([present, rng.choice(present, size=n_keys // 2)]) if duplicates else present.copy()
    rng.shuffle(build)
    n_probe = 400_000
    probe = np.where(rng.random(n_probe) < 0.7, rng.choice(present, size=n_probe), rng.choice(absent, size=n_probe)).astype(dtype)
    pf = oracle.bitmap_from_bools(rng.random(n_probe) < 0.8)
    dp = to_dev(probe, dev)
    results = {}
    for lds in ("1", "0"):
        monkeypatch.setenv("QSX_JOIN_LDS", lds)
        table = capi.JoinTable(key_type, build.size)
        table.build(to_dev(build, dev))
        got = []
        for filt in (None, pf):
            fdev = None if filt is None else bitmap_dev(filt, dev)
            _, rp, rd = oracle_join(oracle, key_type, [build], probe, probe_filter=filt)
            assert int(table.probe_count(dp, filter_bitmap=fdev).item()) == rp.size
            p, b, cnt = table.probe(dp, capacity=rp.size, filter_bitmap=fdev)
            assert int(cnt.item()) == rp.size
            pairs = sorted_pairs(p.cpu().numpy()[:rp.size], b.cpu().numpy()[:rp.size])
            assert np.array_equal(pairs, sorted_pairs(rp, rd))
            want = np.zeros(n_probe, dtype=bool)
            want[rp] = True
            live = np.ones(n_probe, dtype=bool) if filt is None else oracle.bools_from_bitmap(filt, n_probe)
            for anti in (False, True):
                bm, c = table.probe_exists(dp, anti=anti, filter_bitmap=fdev)
                ref = (want != anti) & live
                assert np.array_equal(bitmap_np(bm), oracle.bitmap_from_bools(ref)) and int(c.item()) == int(ref.sum())
            got.append(pairs)
        cuts = [0, 4096, 4096, 123_457, n_probe]
        blocks = [dp[a:b_] for a, b_ in zip(cuts[:-1], cuts[1:])]
        _, rp, rd = oracle_join(oracle, key_type, [build], probe)
        p, b, cnt = table.probe_blocks(blocks, capacity=rp.size)
        assert int(cnt.item()) == rp.size
        assert np.array_equal(sorted_pairs(p.cpu().numpy()[:rp.size], b.cpu().numpy()[:rp.size]), sorted_pairs(rp, rd))
        results[lds] = got
        table.close()
    for a, b in zip(results["1"], results["0"]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("key_type,dtype", [(T.INT, np.int32), (T.LONG, np.int64)])
@pytest.mark.parametrize("table_kind", ["dense", "hashed_dense_keys", "hashed_sparse_keys"])
@pytest.mark.parametrize("filters", [["exact"], ["hash"], ["exact", "hash"], ["anti"], ["exact", "hash", "exact"]])
def test_probe_with_lip_filters_inside_equals_filter_then_probe(capi, oracle, dev, key_type, dtype, table_kind, filters, monkeypatch):
    _probe_lip_case(capi, oracle, dev, key_type, dtype, table_kind, filters, monkeypatch, n_probe=1_300_003)


@pytest.mark.parametrize("key_type,dtype", [(T.INT, np.int32), (T.LONG, np.int64)])
@pytest.mark.parametrize("table_kind", ["dense", "hashed_dense_keys"])
@pytest.mark.parametrize("filters", [["exact"], ["exact", "hash"], ["exact", "hash", "exact"]])
@pytest.mark.parametrize("n_probe,one_pass", [(50_021, None), (3_001, None), (700_001, "0"), (700_001, None)])
def test_probe_lip_whose_probe_takes_the_two_pass_form(capi, oracle, dev, key_type, dtype, table_kind, filters, n_probe, one_pass, monkeypatch):
    """qsx_join_probe_lip's filter-then-probe fallback holds its LIP bitmaps in the call's scratch and
    then runs launch_probe, whose two-pass form (n < 64 Ki rows on a directly addressed table, any n with
    QSX_JOIN_ONE_PASS=0, three filters and n < 1 Mi) reserves scratch on the same stream: the inner reservation must not
    hand out the bytes of the bitmaps the probe is reading."""
    if one_pass is not None:
        monkeypatch.setenv("QSX_JOIN_ONE_PASS", one_pass)
    _probe_lip_case(capi, oracle, dev, key_type, dtype, table_kind, filters, monkeypatch, n_probe=n_probe, plain_filter_legs=False)


def _probe_lip_case(capi, oracle, dev, key_type, dtype, table_kind, filters, monkeypatch, n_probe, plain_filter_legs=True):
    """qsx_join_probe_lip: HashInnerJoinWorkOrder's LIP filters tested inside the probe (one pass over the keys for directly
    addressed tables and shadows; the filter-then-probe sequence inside the call for every other table and for more than two
    filters).  Pairs = the oracle's LIP probes chained into a bitmap + its join under that bitmap: exact, hash and anti
    filters, keys outside an exact filter's range, an input bitmap, duplicate build keys; and the one-pass probe under a
    plain filter (QSX_JOIN_ONE_PASS) against the two-pass form."""
    rng = np.random.default_rng(4242 + len(filters))
    n_build, domain = 60_000, 150_000
    lo = -1000 if dtype == np.int32 else 2**40
    keys = lo + rng.integers(0, domain, size=n_build)
    if table_kind == "hashed_sparse_keys":
        keys = lo + (rng.integers(0, domain, size=n_build) * 7919)
    build = keys.astype(dtype)                                          # duplicates
    span = int(build.max()) - lo + 1
    probe = (lo + rng.integers(-span // 20, span + span // 20, size=n_probe)).astype(dtype)
    if table_kind == "hashed_sparse_keys":
        probe = np.where(rng.random(n_probe) < 0.6, rng.choice(build, size=n_probe), probe).astype(dtype)
    in_filter = oracle.bitmap_from_bools(rng.random(n_probe) < 0.8)
    table = capi.JoinTable(key_type, n_build, key_range=(int(build.min()), int(build.max())) if table_kind == "dense" else None)
    table.build(to_dev(build, dev))
    otable = oracle.JoinTable(key_type, n_build)
    otable.build(build)
    members = build[rng.random(n_build) < 0.5]                           # what the filters were built on
    lips, olips = [], []
    for kind in filters:
        if kind == "hash":
            args = (T.LIP_SINGLE_IDENTITY_HASH, 8 * members.size, 0, False)
        else:      # exact filters cover only part of the probe keys' range: out-of-range keys miss (hit for the anti filter)
            args = (T.LIP_BITVECTOR_EXACT, span, lo, kind == "anti")
        f, of = capi.LipFilter(*args), oracle.LipFilter(*args)
        subset = members[members >= lo] if kind == "hash" else members
        f.build(to_dev(subset.astype(dtype), dev))
        of.build(subset.astype(dtype))
        lips.append(f)
        olips.append(of)
    dp = to_dev(probe, dev)
    for filt in (None, in_filter):
        bitmap = filt
        for of in olips:
            bitmap = of.probe(probe, in_bitmap=bitmap)
        rp, rd = otable.probe(probe, filter_bitmap=bitmap)
        p, b, cnt = table.probe_lip(dp, lips, capacity=rp.size, filter_bitmap=None if filt is None else bitmap_dev(filt, dev))
        assert int(cnt.item()) == rp.size
        assert np.array_equal(sorted_pairs(p.cpu().numpy()[:rp.size], b.cpu().numpy()[:rp.size]), sorted_pairs(rp, rd))
    # the probe under a plain filter: one pass (one reservation per 16 K rows) = two passes (count / scan / write)
    rp, rd = otable.probe(probe, filter_bitmap=in_filter)
    for one_pass in ("1", "0") if plain_filter_legs else ():
        monkeypatch.setenv("QSX_JOIN_ONE_PASS", one_pass)
        p, b, cnt = table.probe(dp, capacity=rp.size, filter_bitmap=bitmap_dev(in_filter, dev))
        assert int(cnt.item()) == rp.size
        assert np.array_equal(sorted_pairs(p.cpu().numpy()[:rp.size], b.cpu().numpy()[:rp.size]), sorted_pairs(rp, rd))
    for f in lips:
        f.close()
    table.close()


def test_first_probes_from_several_streams_wait_for_the_packed_table(capi, dev):
    """A directly addressed table around the size of an XCD's L2 is packed to 3 bytes per key value by its FIRST probe
    (join.hip sealed_pack), on that probe's stream.  Probes that other host threads issue at the same moment on their own
    streams — the Workers' first HashJoin work orders behind a BuildHash (relational_operators/HashJoinOperator.cpp:220-231,
    one work order per probe block) — must run behind the pack kernel whichever way they learn of it: seeing the table
    already marked packed, or finding it packed when they get the mutex they waited for (that second way once let a probe
    read the packed copy while it was being written: half a work order's rows were lost — seen once in some twenty runs of
    twenty concurrent single-block work orders; the window is the first prober's critical section, so this test exercises
    the path without promising to hit it)."""
    import threading
    n_build, n_probe, threads = 1_000_000, 1 << 20, 8
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    build = torch.randperm(n_build, device=dev, generator=g).to(torch.int32)
    probes = [torch.randint(0, n_build, (n_probe,), device=dev, generator=g, dtype=torch.int32) for _ in range(threads)]
    for _ in range(10):
        table = capi.JoinTable(T.INT, n_build, key_range=(0, n_build - 1))
        table.build(build)
        torch.cuda.synchronize()                  # the pipeline breaker between BuildHash and HashJoin
        start = threading.Barrier(threads)
        counts = [None] * threads

        def work(i):
            stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(stream):
                start.wait()
                c = table.probe_count(probes[i])
                stream.synchronize()
                counts[i] = int(c.item())
        pool = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
        for t in pool:
            t.start()
        for t in pool:
            t.join()
        assert counts == [n_probe] * threads, counts
        table.close()


@pytest.mark.parametrize("keys_kind,n_build", [("dense", 1_000_000), ("sparse", 900_000), ("sparse_duplicates", 900_000), ("small", 20_000)])
@pytest.mark.parametrize("verify", ["0", "1"])
def test_first_probes_of_a_hashed_table_from_several_streams(capi, dev, keys_kind, n_build, verify, monkeypatch):
    """The first probe after the builds of a hashed table (join.hip sealed_shadow) no longer waits for the device: the probing
    stream is put behind the event of every stream that cleared or built the table, reads the control words behind them,
    then either copies the entries into a directly addressed shadow (dense keys) or runs the scan that writes the
    fingerprint plane and finds duplicate keys (builds write neither any more) and, for big tables of unique sparse keys, the
    compact plane.  Builds from TWO streams, then eight Worker-like threads probing at once on their own streams — the first
    HashJoin work orders behind a BuildHash (relational_operators/HashJoinOperator.cpp:220-231) — must all see the complete
    table: whoever seals, the others run behind its event.  Ten build / probe rounds per table (clear in between), with
    QSX_JOIN_SEAL_VERIFY=1 (the sealing thread waits for the shadow and reads its error word) and without."""
    import threading
    monkeypatch.setenv("QSX_JOIN_SEAL_VERIFY", verify)
    n_probe, threads = 1 << 20, 8
    g = torch.Generator(device=dev)
    g.manual_seed(17)
    build = torch.randperm(n_build, device=dev, generator=g).to(torch.int32)
    if keys_kind.startswith("sparse"):
        build = (build.long() * 2039 % (2**31 - 1)).to(torch.int32)               # a bijection: unique keys without a dense domain
    per_key = 1
    if keys_kind == "sparse_duplicates":
        build[n_build // 2:] = build[:n_build - n_build // 2]                     # every key twice
        per_key = 2
    half = n_build // 2
    probes = [build[torch.randint(0, n_build, (n_probe,), device=dev, generator=g)] for _ in range(threads)]
    misses = torch.full((n_probe,), 2**31 - 1, dtype=torch.int32, device=dev)      # not a build key
    table = capi.JoinTable(T.INT, n_build)
    side = torch.cuda.Stream(device=dev)
    for _ in range(10):
        table.clear()
        table.build(build[:half], base_tid=0)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            table.build(build[half:], base_tid=half)                               # a second BuildHash work order, on its own stream
        # (no synchronisation here: the pipeline breaker is the builds' calls having returned, not the device being idle)
        start = threading.Barrier(threads)
        counts, missed = [None] * threads, [None] * threads

        def work(i):
            stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(stream):
                start.wait()
                c = table.probe_count(probes[i])
                m = table.probe_count(misses)
                stream.synchronize()
                counts[i], missed[i] = int(c.item()), int(m.item())
        pool = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
        for t in pool:
            t.start()
        for t in pool:
            t.join()
        assert counts == [n_probe * per_key] * threads and missed == [0] * threads, (counts, missed)
        torch.cuda.synchronize()
    table.close()


def _compact_probes(capi):
    import ctypes
    capi.lib.qsx_debug_join_compact_probes.restype = ctypes.c_longlong
    return capi.lib.qsx_debug_join_compact_probes()


@pytest.mark.parametrize("case", ["unique", "duplicates", "large_tuple_ids", "too_few_buckets"])
def test_big_bucketed_tables_are_probed_through_their_compact_plane(capi, oracle, dev, case, monkeypatch):
    """join.hip CompactView: a sealed bucketed table over INT keys without a dense domain, without duplicate keys, with tuple
    ids below 2^24 and more than 65 536 buckets gets a second plane of 4-byte slots {8 identity bits, tuple id} behind
    fingerprints that are identity bits too — bucket + fingerprint + slot ARE the key (the hash is a bijection).  Probes read
    their home bucket there and walk on in the 8-byte table only for keys the build displaced (3-4 % at load 0.8).
    getAllFromValueAccessor's contract (storage/HashTable.hpp:2145-2181): pairs, counts, existence / anti, under a filter and
    over a run of blocks — equal to the oracle, and to the same calls with QSX_JOIN_COMPACT=0.  Tables that must NOT take it
    (duplicate keys, tuple ids of 2^24 and more, too few buckets) answer from the 8-byte slots as before."""
    rng = np.random.default_rng(4100)
    n_build = 900_000 if case != "too_few_buckets" else 700_000
    keys = rng.choice(np.arange(-2**31, 2**31, 2311, dtype=np.int64), size=n_build, replace=False).astype(np.int32)   # sparse: no dense shadow
    if case == "duplicates":
        keys[1000:1010] = keys[:10]
    base_tid = (1 << 24) if case == "large_tuple_ids" else 0
    n_probe = 1_200_007
    probe = np.where(rng.random(n_probe) < 0.6, rng.choice(keys, size=n_probe), rng.integers(-2**31, 2**31, size=n_probe)).astype(np.int32)
    pf = oracle.bitmap_from_bools(rng.random(n_probe) < 0.7)
    ot = oracle.JoinTable(T.INT, n_build)
    ot.build(keys, block_id=0, base_tid=base_tid)
    dp = to_dev(probe, dev)
    answers = {}
    for compact in ("1", "0"):
        monkeypatch.setenv("QSX_JOIN_COMPACT", compact)
        before = _compact_probes(capi)
        table = capi.JoinTable(T.INT, n_build)
        table.build(to_dev(keys, dev), base_tid=base_tid)
        got = []
        for filt in (None, pf):
            fdev = None if filt is None else bitmap_dev(filt, dev)
            rp, rb = ot.probe(probe, filter_bitmap=filt)
            assert int(table.probe_count(dp, filter_bitmap=fdev).item()) == rp.size
            p, b, cnt = table.probe(dp, capacity=rp.size, filter_bitmap=fdev)
            assert int(cnt.item()) == rp.size
            pairs = sorted_pairs(p.cpu().numpy()[:rp.size], b.cpu().numpy()[:rp.size])
            assert np.array_equal(pairs, sorted_pairs(rp, rb))
            want = np.zeros(n_probe, dtype=bool)
            want[rp] = True
            live = np.ones(n_probe, dtype=bool) if filt is None else oracle.bools_from_bitmap(filt, n_probe)
            for anti in (False, True):
                bm, c = table.probe_exists(dp, anti=anti, filter_bitmap=fdev)
                ref = (want != anti) & live
                assert np.array_equal(bitmap_np(bm), oracle.bitmap_from_bools(ref)) and int(c.item()) == int(ref.sum())
            got.append(pairs)
        cuts = [0, 4096, 300_001, n_probe]
        blocks = [dp[a:b_] for a, b_ in zip(cuts[:-1], cuts[1:])]
        rp, rb = ot.probe(probe)
        assert int(table.probe_count_blocks(blocks).item()) == rp.size
        launches = _compact_probes(capi) - before
        assert (launches > 0) == (compact == "1" and case == "unique"), (case, compact, launches)
        answers[compact] = got
        table.close()
    for a, b in zip(answers["1"], answers["0"]):
        assert np.array_equal(a, b)


def test_compact_plane_follows_builds_growth_and_clear(capi, oracle, dev):
    """The compact plane belongs to ONE sealed state of the table: a build behind the first probes drops it (the next probe
    seals again — over a table that has grown by rehash in between, with another bucket count and so other identities), a
    clear empties the table under it.  Every probe equals the oracle's."""
    rng = np.random.default_rng(77)
    universe = np.arange(-2**31, 2**31, 1777, dtype=np.int64)
    keys = rng.choice(universe, size=1_900_000, replace=False).astype(np.int32)
    first, second = keys[:900_000], keys[900_000:]
    probe = np.where(rng.random(800_003) < 0.7, rng.choice(keys, size=800_003), rng.integers(-2**31, 2**31, size=800_003)).astype(np.int32)
    dp = to_dev(probe, dev)
    before = _compact_probes(capi)
    table = capi.JoinTable(T.INT, first.size)          # sized for the first build only: the second one makes it grow
    ot = oracle.JoinTable(T.INT, keys.size)
    table.build(to_dev(first, dev))
    ot.build(first, block_id=0, base_tid=0)

    def same_as_oracle():
        rp, rb = ot.probe(probe)
        p, b, cnt = table.probe(dp, capacity=max(rp.size, 1))
        assert int(cnt.item()) == rp.size
        assert np.array_equal(sorted_pairs(p.cpu().numpy()[:rp.size], b.cpu().numpy()[:rp.size]), sorted_pairs(rp, rb))
    same_as_oracle()
    table.build(to_dev(second, dev), base_tid=first.size)
    ot.build(second, block_id=1, base_tid=first.size)
    same_as_oracle()
    same_as_oracle()                                   # (sealed: straight through the plane)
    assert _compact_probes(capi) - before == 3
    table.clear()
    p, b, cnt = table.probe(dp, capacity=16)
    assert int(cnt.item()) == 0
    table.build(to_dev(second, dev))
    ot2 = oracle.JoinTable(T.INT, second.size)
    ot2.build(second, block_id=0, base_tid=0)
    rp, rb = ot2.probe(probe)
    p, b, cnt = table.probe(dp, capacity=max(rp.size, 1))
    assert int(cnt.item()) == rp.size
    assert np.array_equal(sorted_pairs(p.cpu().numpy()[:rp.size], b.cpu().numpy()[:rp.size]), sorted_pairs(rp, rb))
    table.close()


@pytest.mark.parametrize("flavour", ["dense", "hashed"])
def test_released_tables_hand_their_memory_to_the_next_one(capi, oracle, dev, flavour):
    """qsx_join_table_release (DestroyHashOperator behind the last HashJoin work order): no wait for the device; the table's
    allocations go to the library's idle list and the next table of the same size takes them — its probes must see its own
    keys only."""
    rng = np.random.default_rng(31)
    n = 300_000
    for round_ in range(4):
        build = rng.permutation(n).astype(np.int32) * 3 + round_          # another key set every round
        probe = rng.integers(0, 3 * n + 3, size=500_001).astype(np.int32)
        table = capi.JoinTable(T.INT, n, key_range=(0, 3 * n + 2) if flavour == "dense" else None)
        table.build(to_dev(build, dev))
        _, rp, rb = oracle_join(oracle, T.INT, [build], probe)
        p, b, cnt = table.probe(to_dev(probe, dev), capacity=max(rp.size, 1))
        assert int(cnt.item()) == rp.size
        assert np.array_equal(sorted_pairs(p.cpu().numpy()[:rp.size], b.cpu().numpy()[:rp.size]), sorted_pairs(rp, rb))
        torch.cuda.synchronize()              # what the caller of qsx_join_table_release promises
        table.release()


@pytest.mark.parametrize("flavour", ["dense", "hashed", "hashed_sparse"])
@pytest.mark.parametrize("n", [1_000, 70_001, 1_200_007])
def test_semi_probe_with_lip_filters_inside_equals_the_two_passes(capi, oracle, dev, flavour, n):
    """qsx_join_probe_exists_lip: the existence bitmap and count of qsx_lip_probe followed by qsx_join_probe_exists — one or two
    LIP filters, with and without an input bitmap, directly addressed tables (the fused kernel from 64 Ki rows), hashed tables
    with and without a directly addressed shadow (the two passes inside the call)."""
    rng = np.random.default_rng(n + len(flavour))
    spread = 1 if flavour != "hashed_sparse" else 1_000
    build = (rng.permutation(40_000)[:25_000] * spread).astype(np.int32)
    probe = (rng.integers(0, 50_000, size=n) * spread).astype(np.int32)
    table = capi.JoinTable(T.INT, build.size, key_range=(0, 39_999) if flavour == "dense" else None)
    table.build(to_dev(build, dev))
    lip_a = capi.LipFilter(T.LIP_BITVECTOR_EXACT, 50_000 * spread, min_value=0)
    lip_a.build(to_dev(build[build % 3 != 0], dev))
    lip_b = capi.LipFilter(T.LIP_SINGLE_IDENTITY_HASH, 8_191, min_value=0)
    lip_b.build(to_dev(build[build % 5 != 0], dev))
    d_probe = to_dev(probe, dev)
    filt = bitmap_dev(oracle.bitmap_from_bools(rng.random(n) < 0.7), dev)
    for lips in ([lip_a], [lip_a, lip_b], []):
        for f in (None, filt):
            got, cnt = table.probe_exists_lip(d_probe, lips, filter_bitmap=f)
            cur = f
            for lip in lips:
                cur = lip.probe(d_probe, in_bitmap=cur)[0]
            want, want_cnt = table.probe_exists(d_probe, filter_bitmap=cur)
            assert torch.equal(got, want), (flavour, n, len(lips), f is not None)
            assert int(cnt.item()) == int(want_cnt.item())
