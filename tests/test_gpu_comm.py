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
