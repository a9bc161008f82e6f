"""Worker of tests/test_distributed_gloo.py: one rank of a 2-rank gloo job that runs the
multi-GPU rank logic of quickstep_amd/distributed.py on CPU tensors, with the CPU checker
plugged in as `ops` (tests may use the oracle; the product passes quickstep_amd.capi)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle_ops import OracleOps  # noqa: E402
from quickstep_amd import distributed as qd  # noqa: E402
from quickstep_amd import plans  # noqa: E402
from quickstep_amd import types as T  # noqa: E402


class FakeAggState:
    """Dense per-group image [count, sum] over 16 groups: enough to exercise all-gather + merge order."""
    device = None

    def __init__(self, values):
        self.image = torch.zeros(32, dtype=torch.int64)
        for g, v in values:
            self.image[g] += 1
            self.image[16 + g] += v

    def export(self, device):
        return self.image.clone()

    def import_merge(self, other):
        self.image += other


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo")
    out_dir = sys.argv[1]
    n_build, n_probe = 5000, 40000
    rng = np.random.default_rng(100 + rank)
    # every rank holds a slice of both relations; keys collide across ranks and include negatives
    build_keys = rng.integers(-2000, 2000, size=n_build).astype(np.int32)
    probe_keys = rng.integers(-2500, 2500, size=n_probe).astype(np.int32)
    join = qd.PartitionedHashJoin(OracleOps, T.INT, n_build * 2)
    join.build(torch.from_numpy(build_keys), rank * n_build)
    probe_tids, build_tids, op, ob, cnt = join.probe(torch.from_numpy(probe_keys), rank * n_probe)
    gp, gb = join.materialize(probe_tids, build_tids, op, ob, cnt)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), build_keys=build_keys, probe_keys=probe_keys,
             pairs_probe=gp.numpy(), pairs_build=gb.numpy())

    # the same with exact statistics of a dense, non-negative key domain: every rank's table is the
    # strided directly addressed flavour; the wrapper asserts the progression precondition
    dom = (3, 4001)
    bk = rng.integers(dom[0], dom[1] + 1, size=n_build).astype(np.int32)
    pk = rng.integers(0, 4500, size=n_probe).astype(np.int32)
    join2 = qd.PartitionedHashJoin(OracleOps, T.INT, n_build * 2, key_domain=dom)
    assert join2.table.key_range is not None and join2.table.key_stride == world
    join2.build(torch.from_numpy(bk), rank * n_build)
    pt2, bt2, op2, ob2, cnt2 = join2.probe(torch.from_numpy(pk), rank * n_probe)
    gp2, gb2 = join2.materialize(pt2, bt2, op2, ob2, cnt2)
    np.savez(os.path.join(out_dir, f"dense_rank{rank}.npz"), build_keys=bk, probe_keys=pk,
             pairs_probe=gp2.numpy(), pairs_build=gb2.numpy())
    assert qd.rank_progression((0, 10), 3, 1) is None and qd.rank_progression((-4, 10), 2, 1) is None
    assert qd.rank_progression((5, 6), 4, 0) is None and qd.rank_progression((5, 17), 4, 2) == (6, 14)

    # broadcast join (small build side): all-gather of the build rows, local probe, no probe shuffle
    join3 = qd.BroadcastHashJoin(OracleOps, T.INT, n_build * world)
    assert join3.build(torch.from_numpy(build_keys), rank * n_build) == n_build * world
    _, _, op3, ob3, cnt3 = join3.probe(torch.from_numpy(probe_keys), rank * n_probe)
    gp3, gb3 = join3.materialize(None, None, op3, ob3, cnt3)
    np.savez(os.path.join(out_dir, f"bcast_rank{rank}.npz"), build_keys=build_keys, probe_keys=probe_keys,
             pairs_probe=gp3.numpy(), pairs_build=gb3.numpy())

    # uneven shares (the general path: padded all-gather, one build per source rank with its own tid base)
    mine = build_keys[: n_build - 37 * rank]
    join4 = qd.BroadcastHashJoin(OracleOps, T.INT, n_build * world)
    assert join4.build(torch.from_numpy(mine), rank * n_build) == sum(n_build - 37 * r for r in range(world))
    _, _, op4, ob4, cnt4 = join4.probe(torch.from_numpy(probe_keys), rank * n_probe)
    gp4, gb4 = join4.materialize(None, None, op4, ob4, cnt4)
    np.savez(os.path.join(out_dir, f"bcast_uneven_rank{rank}.npz"), build_keys=mine, probe_keys=probe_keys,
             pairs_probe=gp4.numpy(), pairs_build=gb4.numpy())

    # partial aggregate merge
    vals = [(int(g), int(v)) for g, v in zip(rng.integers(0, 16, size=1000), rng.integers(0, 100, size=1000))]
    st = FakeAggState(vals)
    local = st.image.clone()
    qd.merge_agg_state_images(OracleOps, st)
    total = local.clone()
    dist.all_reduce(total)
    assert torch.equal(st.image, total)

    # dense image all-reduce: existence bits OR, integer SUM, f64 SUM
    exist_words, entries = 2, 5
    img = torch.zeros(exist_words + 2 * entries, dtype=torch.int64)
    img[0] = 1 << rank
    img[1] = 0b1010 if rank == 0 else 0b0110
    img[exist_words:exist_words + entries] = torch.arange(entries) + rank
    img[exist_words + entries:] = (torch.arange(entries, dtype=torch.float64) * 0.5 + rank).view(torch.int64)
    qd.allreduce_dense_agg_image(img, exist_words, entries, int_col_mask=0b01, num_cols=2)
    assert int(img[0]) == (1 << world) - 1 and int(img[1]) == 0b1110
    assert img[exist_words:exist_words + entries].tolist() == [world * i + sum(range(world)) for i in range(entries)]
    assert np.allclose(img[exist_words + entries:].view(torch.float64).numpy(),
                       np.arange(entries) * 0.5 * world + sum(range(world)))
    # the bit-OR as RCCL has to do it (no bitwise reductions there): all-gather + local OR
    real_backend = dist.get_backend
    dist.get_backend = lambda group=None: "nccl"
    try:
        bits = torch.tensor([1 << rank, 0b1010 if rank == 0 else 0b0110, -1 if rank == 1 else 0], dtype=torch.int64)
        qd._allreduce_or(bits)
        assert bits.tolist() == [(1 << world) - 1, 0b1110, -1]
    finally:
        dist.get_backend = real_backend
    # dense image reduce-scatter: rank r ends up with the merged keys of finalize partition r only
    entries = 150                                   # not a multiple of 64 * world: partitions share boundary words
    exist_words = (entries + 63) // 64
    r2 = np.random.default_rng(7 + rank)
    present = r2.random(entries) < 0.4
    cnt = np.where(present, r2.integers(1, 5, size=entries), 0).astype(np.int64)
    sm = np.where(present, r2.normal(size=entries), 0.0)
    mn = np.where(present, r2.integers(-50, 50, size=entries), np.iinfo(np.int64).max).astype(np.int64)
    words = np.zeros(exist_words, dtype=np.uint64)
    for k in np.nonzero(present)[0]:
        words[k >> 6] |= np.uint64(1) << np.uint64(k & 63)
    img = torch.from_numpy(np.concatenate([words.view(np.int64), cnt, sm.view(np.int64), mn]))
    got = qd.reduce_scatter_dense_agg_image(img.clone(), exist_words, entries, int_col_mask=0b001, num_cols=3,
                                            min_max_cols={2: "min"})
    # what the all-reduce of the same images holds, restricted to this rank's partition
    ref = img.clone()
    qd.allreduce_dense_agg_image(ref, exist_words, entries, int_col_mask=0b001, num_cols=3, min_max_cols={2: "min"})
    begin, end = qd.dense_partition_range(entries, world, rank)
    ref_bits = [(int(ref[k >> 6]) >> (k & 63)) & 1 for k in range(entries)]
    got_bits = [(int(got[k >> 6]) >> (k & 63)) & 1 for k in range(entries)]
    assert got_bits == [b if begin <= k < end else 0 for k, b in enumerate(ref_bits)]
    for col in range(3):
        a = got[exist_words + col * entries: exist_words + (col + 1) * entries]
        b = ref[exist_words + col * entries: exist_words + (col + 1) * entries]
        if col == 1:
            assert np.allclose(a.view(torch.float64)[begin:end].numpy(), b.view(torch.float64)[begin:end].numpy(), rtol=1e-12)
            assert float(a.view(torch.float64)[:begin].abs().sum() + a.view(torch.float64)[end:].abs().sum()) == 0.0
        else:
            assert torch.equal(a[begin:end], b[begin:end])
            outside = torch.cat([a[:begin], a[end:]])
            assert bool((outside == (np.iinfo(np.int64).max if col == 2 else 0)).all())
    # BASELINE config 4: partitioned join with one 8-byte payload column per side (strided directly addressed tables)
    cpu = torch.device("cpu")
    orders_per_rank = 3000
    c4 = plans.generate_c4_inputs(cpu, orders_per_rank, rank)
    pj = plans.PartitionedJoin(OracleOps, orders_per_rank * world, orders_per_rank, dense=True)
    cols, moved = pj.step(c4, rank * orders_per_rank, 0)
    assert plans.PartitionedJoin.check(cols) and bool(((cols[0] & (world - 1)) == rank).all())
    np.savez(os.path.join(out_dir, f"c4_rank{rank}.npz"), l_key=c4["l_orderkey"].numpy(), l_pay=c4["l_payload"].numpy(),
             out_key=cols[0].numpy(), out_o=cols[1].numpy(), out_l=cols[2].numpy())

    # BASELINE config 5: the distributed Q3 plan (LIP OR, broadcast builds, dense-state reduce-scatter, global top 10)
    q3_in = plans.generate_q3_inputs(cpu, 0.004, rank, world)
    saved = {"in_" + k: v.numpy() for k, v in q3_in.items() if torch.is_tensor(v)}
    for fused in (True, False):
        q3 = plans.DistributedQ3(OracleOps, q3_in["customers_total"], q3_in["orders_total"], use_lip=True, fused=fused)
        res = q3.run(q3_in, tid_base_orders=rank * q3_in["o_orderkey"].numel())
        keys_r, vals_r, _, groups_r = q3.state.finalize(cpu, partition=rank, num_partitions=world)
        tag = "f" if fused else "g"
        saved.update({f"{tag}_keys": keys_r[0].numpy(), f"{tag}_rev": vals_r[0].numpy(), f"{tag}_pairs": np.int64(res["pairs"]),
                      f"{tag}_top_keys": res["top_keys"].numpy(), f"{tag}_top_rev": res["top_revenue"].numpy()})
        assert q3.comm_bytes > 0
    np.savez(os.path.join(out_dir, f"q3_rank{rank}.npz"), **saved)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
