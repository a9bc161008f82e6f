"""N = 2, 3, 4 and 8 with the product's kernels: the rank processes share cuda:0 (tests/dist_worker_gpu.py), ops =
quickstep_amd.capi.  Transports: "gloo" = torch.distributed collectives with host staging; "capi" = the C ABI's own
multi-GPU entry points (qsx_alltoallv, qsx_allgather, qsx_bitmap_allreduce_or, qsx_agg_reduce_scatter,
qsx_agg_allgather_merge) over the tests' loopback stand-in for RCCL, at world 2, at world 3 (not a power of two:
hash % P partitions, hashed tables, key ranges that do not divide), at world 4 and at world 8 — the partition count of the
node this is built for (pid = h & 7, catalog/PartitionSchemeHeader.hpp:207-214).  The union of what the ranks produce must be what one
process / the CPU oracle produces: shuffle joins (hashed and strided directly addressed tables, payload columns =
BASELINE config 4), broadcast join, Q1 state merge, dense state reduce-scatter, and the distributed Q3 plan (BASELINE
config 5)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from quickstep_amd import types as T
from helpers import q3_reference_numpy, sorted_pairs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOOPBACK = os.path.join(ROOT, "tests", "cpp", "bin", "libloopback_rccl.so")


def launch_ranks(world, script, args, extra_env=None, timeout=900):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", script)] + [str(a) for a in args]
    env = dict(os.environ, OMP_NUM_THREADS="1", QSX_AGG_JIT="0")
    env.update(extra_env or {})
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0 and r.stdout.count("RANKS_OK") == world, r.stdout[-3000:] + r.stderr[-6000:]


@pytest.fixture(scope="module", params=[(2, "gloo"), (2, "capi"), (3, "capi"), (4, "capi"), (8, "capi")], ids=lambda p: f"world{p[0]}-{p[1]}")
def ranks(request, tmp_path_factory):
    world, transport = request.param
    out = tmp_path_factory.mktemp(f"ranks_{world}_{transport}")
    extra = {}
    if transport == "capi":
        assert os.path.exists(LOOPBACK), "tests/cpp/bin/libloopback_rccl.so is not built (make -C quickstep_amd/host)"
        extra["QSX_RCCL_LIBRARY"] = LOOPBACK
        extra["QSX_ALLOW_TEST_TRANSPORT"] = "1"
    launch_ranks(world, "dist_worker_gpu.py", [out, transport], extra)
    loaded = [np.load(out / f"rank{i}.npz") for i in range(world)]
    assert all(int(d["world_size_seen"]) == world for d in loaded)        # what the communicator itself reported on every rank
    return loaded


def test_shuffle_join_hashed_tables_equals_oracle_join(ranks, oracle):
    build = np.concatenate([d["h_build"] for d in ranks])            # global tid = rank * n + local row
    probe = np.concatenate([d["h_probe"] for d in ranks])
    got = np.concatenate([np.stack([d["h_pp"], d["h_pb"]], 1) for d in ranks])
    for i, d in enumerate(ranks):                                     # a pair sits on the rank that owns its key's partition
        assert ((probe[d["h_pp"]].astype(np.uint32) % len(ranks)) == i).all()       # hash = the zero-extended key, pid = h % P
    t = oracle.JoinTable(T.INT, build.size)
    t.build(build)
    p, b = t.probe(probe)
    assert np.array_equal(sorted_pairs(got[:, 0], got[:, 1]), sorted_pairs(p, b))


def test_broadcast_join_equals_oracle_join(ranks, oracle):
    build = np.concatenate([d["h_build"] for d in ranks])
    probe = np.concatenate([d["h_probe"] for d in ranks])
    got = np.concatenate([np.stack([d["b_pp"], d["b_pb"]], 1) for d in ranks])
    n_probe = ranks[0]["h_probe"].size
    for i, d in enumerate(ranks):                                     # produced by the rank that holds the probe row
        assert ((d["b_pp"] // n_probe) == i).all()
    t = oracle.JoinTable(T.INT, build.size)
    t.build(build)
    p, b = t.probe(probe)
    assert np.array_equal(sorted_pairs(got[:, 0], got[:, 1]), sorted_pairs(p, b))


def test_config4_partitioned_join_with_payloads(ranks):
    """Output relation (key, o_payload, l_payload): exactly one row per lineitem row, payloads of the right rows."""
    l_key = np.concatenate([d["c4_l_key"] for d in ranks])
    l_pay = np.concatenate([d["c4_l_pay"] for d in ranks])
    out_key = np.concatenate([d["c4_out_key"] for d in ranks])
    out_o = np.concatenate([d["c4_out_o"] for d in ranks])
    out_l = np.concatenate([d["c4_out_l"] for d in ranks])
    assert out_key.size == l_key.size                                  # every line has exactly one order
    assert np.array_equal(out_o, out_key.astype(np.int64) * 3 + 1)
    # l_payload identifies the lineitem row: the multiset of (key, l_payload) is that of the input
    a = np.stack([out_key.astype(np.int64), out_l], 1)
    b = np.stack([l_key.astype(np.int64), l_pay], 1)
    assert np.array_equal(a[np.lexsort((a[:, 1], a[:, 0]))], b[np.lexsort((b[:, 1], b[:, 0]))])
    # (key 4 B + tid 4 B + payload 8 B) of both relations went through the exchange
    for d in ranks:
        assert int(d["c4_moved"]) > 0


def test_q1_states_merged_across_ranks_equal_the_single_state(ranks, oracle):
    from dist_worker_gpu import q1_config
    cfg = q1_config()
    cols = [np.concatenate([d[f"q1_col{i}"] for d in ranks]) for i in range(6)]
    ost = oracle.AggState(cfg)
    ost.update(cols, cols[0].size)
    rkeys, rvals, _ = ost.finalize()
    ref_order = np.lexsort([rkeys[1], rkeys[0]])
    for d in ranks:                                                    # every rank holds the whole merged table
        order = np.lexsort([d["q1_key1"], d["q1_key0"]])
        assert np.array_equal(d["q1_key0"][order], rkeys[0][ref_order]) and np.array_equal(d["q1_key1"][order], rkeys[1][ref_order])
        for a in range(cfg.num_aggs):
            gv, rv = d[f"q1_val{a}"][order], rvals[a][ref_order]
            if gv.dtype == np.int64:
                assert np.array_equal(gv, rv)
            else:
                assert np.allclose(gv, rv, rtol=1e-6, atol=0.0)


def test_dense_state_reduce_scatter_gives_every_rank_its_key_range(ranks):
    keys = np.concatenate([d["d_keys_in"] for d in ranks])
    vals = np.concatenate([d["d_vals_in"] for d in ranks])
    entries = 5_003
    cnt = np.bincount(keys, minlength=entries)
    sm = np.bincount(keys, weights=vals, minlength=entries)
    mn = np.full(entries, np.inf)
    np.minimum.at(mn, keys, vals)
    world = len(ranks)
    length = (entries + world - 1) // world
    seen = []
    for r, d in enumerate(ranks):
        k = d["d_key"]
        assert ((k >= r * length) & (k < min((r + 1) * length, entries))).all()
        assert np.array_equal(d["d_cnt"], cnt[k])
        assert np.allclose(d["d_sum"], sm[k], rtol=1e-9, atol=1e-12)
        assert np.array_equal(d["d_min"], mn[k])
        seen.append(k)
    seen = np.sort(np.concatenate(seen))
    assert np.array_equal(seen, np.nonzero(cnt)[0])


@pytest.mark.parametrize("tag", ["f", "g"])
def test_config5_q3_pipeline_two_ranks(ranks, tag):
    inputs = [{k[len("q3in_"):]: d[k] for k in d.files if k.startswith("q3in_")} for d in ranks]
    keys, sums, pairs = q3_reference_numpy(inputs)
    assert sum(int(d[f"q3{tag}_pairs"]) for d in ranks) == pairs
    got_keys = np.concatenate([d[f"q3{tag}_keys"] for d in ranks])
    got_rev = np.concatenate([d[f"q3{tag}_rev"] for d in ranks])
    order = np.argsort(got_keys)
    assert np.array_equal(got_keys[order], keys)                       # integer keys: exact
    assert np.allclose(got_rev[order], sums, rtol=1e-6, atol=0.0)      # DOUBLE sums: 1e-6 relative (north_star)
    top = np.argsort(-sums, kind="stable")[:10]
    for d in ranks:                                                    # every rank ends with the same global top 10
        assert np.allclose(d[f"q3{tag}_top_rev"], sums[top], rtol=1e-6, atol=0.0)
        assert np.array_equal(np.sort(d[f"q3{tag}_top_keys"]), np.sort(keys[top]))
