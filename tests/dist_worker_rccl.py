"""Worker of tests/test_gpu_partitioned_join.py::test_distributed_surface_over_rccl: ONE rank over the "nccl" backend
(= RCCL) on the GPU box, running every collective quickstep_amd/distributed.py issues with the product's own ops
(quickstep_amd.capi).  A 1-GPU box cannot exchange anything, but the backend still has to accept every dtype / reduce-op
/ split-size combination the multi-GPU path uses — RCCL rejects e.g. bitwise reductions — and the results of a
single-rank job are the local results, which are checked."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import quickstep_amd.capi as capi  # noqa: E402
from quickstep_amd import distributed as qd  # noqa: E402
from quickstep_amd import types as T  # noqa: E402


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group(backend="nccl", device_id=dev)
    world, rank = dist.get_world_size(), dist.get_rank()
    assert world == 1
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    n_build, n_probe = 50_000, 400_000
    build = torch.randperm(n_build, device=dev, generator=g, dtype=torch.int32)
    probe = torch.randint(0, n_build, (n_probe,), device=dev, generator=g, dtype=torch.int32)
    for join in (qd.PartitionedHashJoin(capi, T.INT, n_build, key_domain=(0, n_build - 1)), qd.PartitionedHashJoin(capi, T.INT, n_build),
                 qd.BroadcastHashJoin(capi, T.INT, n_build, key_domain=(0, n_build - 1))):
        assert join.build(build, 0) == n_build
        pt, bt, op, ob, cnt = join.probe(probe, 0, capacity=n_probe)
        gp, gb = join.materialize(pt, bt, op, ob, cnt)
        assert gp.numel() == n_probe and bool((build[gb.long()] == probe[gp.long()]).all())

    # hash-state merge (all-gather of images + import-merge; one rank: nothing to merge, the collectives still run)
    keys = torch.randint(0, 50, (100_000,), device=dev, generator=g, dtype=torch.int32)
    vals = torch.rand(100_000, device=dev, generator=g, dtype=torch.float64)
    cfg = T.make_agg_config(T.AGG_GENERIC, [(T.INT, None), (T.DOUBLE, None)], keys=[0],
                            aggs=[(T.AGG_SUM, T.col(1)), (T.AGG_COUNT_STAR, None), (T.AGG_MIN, T.col(1))], est_groups=64)
    st = capi.AggState(cfg)
    st.update([keys, vals], keys.numel())
    before = st.export(dev).clone()
    qd.merge_agg_state_images(capi, st)
    assert torch.equal(st.export(dev), before)

    # dense state: reduce-scatter and all-reduce of the image (SUM int64, SUM f64, MIN int64, existence OR)
    entries = 1000
    dcfg = T.make_agg_config(T.AGG_COLLISION_FREE, [(T.INT, None), (T.DOUBLE, None)], keys=[0],
                             aggs=[(T.AGG_COUNT_STAR, None), (T.AGG_SUM, T.col(1)), (T.AGG_MIN, T.col(1))], num_entries=entries)
    ds = capi.AggState(dcfg)
    dkeys = torch.randint(0, entries, (200_000,), device=dev, generator=g, dtype=torch.int32)
    ds.update([dkeys, vals.repeat(2)], dkeys.numel())
    want = [x.cpu().numpy() for x in ds.finalize(dev)[1]]
    image = ds.export(dev)
    exist_words = (entries + 63) // 64
    num_cols = (image.numel() - exist_words) // entries
    assert num_cols == 3
    reduced = qd.reduce_scatter_dense_agg_image(image.clone(), exist_words, entries, int_col_mask=0b001, num_cols=3, min_max_cols={2: "min"})
    ds.clear()
    ds.import_merge(reduced)
    got = [x.cpu().numpy() for x in ds.finalize(dev, partition=rank, num_partitions=world)[1]]
    for a, b in zip(got, want):
        assert np.allclose(a, b, rtol=1e-12)
    both = image.clone()
    qd.allreduce_dense_agg_image(both, exist_words, entries, int_col_mask=0b001, num_cols=3, min_max_cols={2: "min"})
    assert torch.equal(both, image)
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_SURFACE_OK")


if __name__ == "__main__":
    main()
