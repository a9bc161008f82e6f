"""The multi-GPU entry points of the C ABI (qsx_comm_*, qsx_alltoallv, qsx_allgather, qsx_bitmap_allreduce_or,
qsx_agg_reduce_scatter, qsx_agg_allgather_merge: RCCL bound at run time inside libqsx.so) with ONE rank on the GPU box: the
library must find RCCL, create a communicator, and every collective must run and return the local data — a dtype or call
pattern RCCL rejects shows up here.  (Two ranks on one GPU are refused by RCCL; the rank logic at N = 2 with the product's
kernels is tests/test_gpu_two_ranks.py.)"""
import numpy as np
import pytest
import torch

from quickstep_amd import types as T

pytestmark = pytest.mark.gpu


def test_single_rank_collectives_through_the_c_abi(capi, dev):
    comm = capi.Comm(1, 0, capi.Comm.unique_id())
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    counts = torch.tensor([12345], dtype=torch.int64, device=dev)
    assert capi.Comm.exchange_counts(comm, counts).tolist() == [12345]
    for dtype in (torch.int32, torch.int64, torch.float64, torch.uint8):
        col = torch.randint(0, 100, (12345,), device=dev, generator=g).to(dtype)
        got = comm.alltoallv(col, [12345], [12345])
        torch.cuda.synchronize()
        assert torch.equal(got, col)
        assert torch.equal(comm.allgather(col), col)
    words = torch.randint(-2**62, 2**62, (1000,), device=dev, generator=g, dtype=torch.int64)
    before = words.clone()
    comm.bitmap_allreduce_or(words)
    torch.cuda.synchronize()
    assert torch.equal(words, before)
    # states: one rank has nothing to merge, the entry points still validate the state kind
    keys = torch.randint(0, 500, (100_000,), device=dev, generator=g, dtype=torch.int32)
    vals = torch.rand(100_000, device=dev, generator=g, dtype=torch.float64)
    dense = capi.AggState(T.make_agg_config(T.AGG_COLLISION_FREE, [(T.INT, None), (T.DOUBLE, None)], keys=[0],
                                             aggs=[(T.AGG_COUNT_STAR, None), (T.AGG_SUM, T.col(1)), (T.AGG_MIN, T.col(1))], num_entries=500))
    hashed = capi.AggState(T.make_agg_config(T.AGG_GENERIC, [(T.INT, None), (T.DOUBLE, None)], keys=[0],
                                              aggs=[(T.AGG_SUM, T.col(1))], est_groups=64))
    for st in (dense, hashed):
        st.update([keys, vals], keys.numel())
    want = [x.clone() for x in dense.finalize(dev)[1]]
    comm.agg_reduce_scatter(dense)
    comm.agg_allgather_merge(hashed)
    for a, b in zip(dense.finalize(dev, partition=0, num_partitions=1)[1], want):
        assert torch.equal(a, b)
    with pytest.raises(capi.QsxError):
        comm.agg_reduce_scatter(hashed)
    with pytest.raises(capi.QsxError):
        comm.agg_allgather_merge(dense)
    comm.close()


# ---- world 2, 3, 4 and 8 over the loopback transport ------------------------------------------------------------------
@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_c_abi_collectives_at_world_2_to_8(world, tmp_path):
    """The N > 1 branches of qsx_exchange_counts / qsx_alltoallv / qsx_allgather / qsx_bitmap_allreduce_or /
    qsx_agg_reduce_scatter / qsx_agg_allgather_merge (tests/comm_loopback_worker.py: rank processes sharing cuda:0, libqsx.so
    bound to tests/cpp/bin/libloopback_rccl.so).  The workers assert bit-equality with quickstep_amd/distributed.py's
    torch.distributed route; here the union of the ranks' results is checked against numpy restatements of the aggregates."""
    import functools
    from test_gpu_two_ranks import LOOPBACK, launch_ranks
    from comm_loopback_worker import DENSE_CASES, dense_inputs, hash_inputs
    import os
    assert os.path.exists(LOOPBACK), "tests/cpp/bin/libloopback_rccl.so is not built (make -C quickstep_amd/host)"
    launch_ranks(world, "comm_loopback_worker.py", [tmp_path], {"QSX_RCCL_LIBRARY": LOOPBACK, "QSX_ALLOW_TEST_TRANSPORT": "1"})
    ranks = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]

    want = functools.reduce(np.bitwise_or, [d["or_in"] for d in ranks])
    for d in ranks:
        assert np.array_equal(d["or_out"], want)

    def aggregate(kind, keys, values, entries):
        """numpy restatement per key: (values of present keys, present mask)"""
        present = np.bincount(keys, minlength=entries) > 0
        if kind == "count":
            return np.bincount(keys, minlength=entries), present
        if kind == "sum":
            if values.dtype == np.float64:
                return np.bincount(keys, weights=values, minlength=entries), present     # multiples of 1/64: exact
            out = np.zeros(entries, dtype=np.int64)
            np.add.at(out, keys, values)
            return out, present
        out = np.full(entries, values.max() if kind == "min" else values.min(), dtype=values.dtype)
        (np.minimum if kind == "min" else np.maximum).at(out, keys, values)
        return out, present

    for name, (entries, _, aggs) in DENSE_CASES.items():
        inputs = [dense_inputs(name, r) for r in range(world)]
        keys = np.concatenate([i[0] for i in inputs])
        cols = {1: np.concatenate([i[1] for i in inputs]), 2: np.concatenate([i[2] for i in inputs])}
        length = (entries + world - 1) // world
        seen = []
        for r, d in enumerate(ranks):
            k = d[f"dense_{name}_key"]
            assert ((k >= min(r * length, entries)) & (k < min((r + 1) * length, entries))).all(), name   # only its own key range
            assert (np.diff(k) > 0).all()
            seen.append(k)
            for i, (kind, arg) in enumerate(aggs):
                got = d[f"dense_{name}_val{i}"]
                if kind == T.AGG_COUNT_STAR:
                    ref, _ = aggregate("count", keys, None, entries)
                elif kind == T.AGG_AVG:
                    s_, _ = aggregate("sum", keys, cols[arg.index], entries)
                    c_, _ = aggregate("count", keys, None, entries)
                    ref = s_ / np.maximum(c_, 1)
                else:
                    ref, _ = aggregate({T.AGG_SUM: "sum", T.AGG_MIN: "min", T.AGG_MAX: "max"}[kind], keys, cols[arg.index], entries)
                assert np.array_equal(got, ref[k].astype(got.dtype)), (name, i)
                assert not d[f"dense_{name}_null{i}"].any()
        assert np.array_equal(np.concatenate(seen), np.nonzero(np.bincount(keys, minlength=entries))[0]), name   # every group exactly once

    inputs = [hash_inputs(r) for r in range(world)]
    keys = np.concatenate([i[0] for i in inputs])
    a = np.concatenate([i[1] for i in inputs])
    b = np.concatenate([i[2] for i in inputs])
    uniq, inv = np.unique(keys, return_inverse=True)
    sums = np.bincount(inv, weights=a)
    counts = np.bincount(inv)
    mins = np.full(uniq.size, b.max())
    np.minimum.at(mins, inv, b)
    maxs = np.full(uniq.size, a.min())
    np.maximum.at(maxs, inv, a)
    for d in ranks:                                  # every rank ends with the whole merged table
        assert np.array_equal(d["hash_key"], uniq)
        assert np.array_equal(d["hash_val0"], sums) and np.array_equal(d["hash_val1"], counts)
        assert np.array_equal(d["hash_val2"], mins) and np.array_equal(d["hash_val3"], maxs)


def test_the_test_transport_is_refused_without_its_switch():
    """QSX_RCCL_LIBRARY alone must not put another library under the collectives (csrc/comm.hip): a process that carries it
    without QSX_ALLOW_TEST_TRANSPORT=1 gets no transport at all — QSX_ERR_COMM — instead of the substitute or a silent fall-back."""
    import os
    import subprocess
    import sys
    from test_gpu_two_ranks import LOOPBACK
    code = ("import quickstep_amd.capi as capi\n"
            "try:\n    capi.Comm.unique_id()\nexcept capi.QsxError as e:\n    print('refused:', e)\nelse:\n    print('bound')\n")
    env = dict(os.environ, QSX_RCCL_LIBRARY=LOOPBACK)
    env.pop("QSX_ALLOW_TEST_TRANSPORT", None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert "refused:" in r.stdout and "QSX_ALLOW_TEST_TRANSPORT" in r.stdout, r.stdout + r.stderr
