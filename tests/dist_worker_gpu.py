"""Worker of tests/test_gpu_two_ranks.py: one of the rank processes (2 or 3) that share the box's single GPU and run the
multi-GPU rank logic (quickstep_amd/distributed.py, quickstep_amd/plans.py) with the PRODUCT's kernels (ops =
quickstep_amd.capi).  RCCL refuses two ranks on one device, so argv[2] picks the transport:
  gloo  the process group is gloo and distributed.py stages the device tensors through the host (its _Transport) — same
        bytes, same order, same kernels on either side of every exchange;
  capi  every exchange goes through the C ABI's multi-GPU entry points (qd.CapiGroup -> qsx_alltoallv, qsx_allgather,
        qsx_bitmap_allreduce_or, qsx_agg_reduce_scatter, qsx_agg_allgather_merge), libqsx.so bound to the tests' loopback
        transport instead of RCCL (QSX_RCCL_LIBRARY = tests/cpp/bin/libloopback_rccl.so).
Every rank writes what it produced to argv[1]; the parent test compares the union with the CPU oracle."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import quickstep_amd.capi as capi  # noqa: E402
from quickstep_amd import distributed as qd  # noqa: E402
from quickstep_amd import plans  # noqa: E402
from quickstep_amd import types as T  # noqa: E402


def q1_config():
    return T.make_agg_config(
        T.AGG_COMPACT_KEY,
        columns=[(T.CHAR, 1), (T.CHAR, 1), (T.DOUBLE, None), (T.DOUBLE, None), (T.DOUBLE, None), (T.DOUBLE, None)],
        keys=[0, 1],
        instrs=[(T.EX_SUB, 0, T.const(0), T.col(4)), (T.EX_MUL, 1, T.col(3), T.temp(0)),
                (T.EX_ADD, 2, T.const(0), T.col(5)), (T.EX_MUL, 3, T.temp(1), T.temp(2))],
        consts=[1.0],
        aggs=[(T.AGG_SUM, T.col(2)), (T.AGG_SUM, T.col(3)), (T.AGG_SUM, T.temp(1)), (T.AGG_SUM, T.temp(3)),
              (T.AGG_AVG, T.col(4)), (T.AGG_COUNT_STAR, None)],
        est_groups=6)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out_dir, transport = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "gloo"
    torch.cuda.set_device(0)                      # all ranks: the one GPU of the box
    dev = torch.device("cuda", 0)
    assert capi.device_count() >= 1
    dist.init_process_group(backend="gloo")
    group = None
    if transport == "capi":
        assert os.environ.get("QSX_RCCL_LIBRARY"), "the capi transport of this test needs the loopback library"
        group = qd.CapiGroup.from_torch_group(capi, dev)
        assert (group.world, group.rank) == (world, rank)
    rng = np.random.default_rng(500 + rank)
    seen = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(seen)
    save = {"world_size_seen": np.int64(group.world if group is not None else int(seen.item()))}
    assert int(seen.item()) == world

    # ---- shuffle join, hashed tables: keys collide across ranks, negatives included, duplicates on the build side
    n_build, n_probe = 30_000, 250_000
    bk = rng.integers(-20_000, 20_000, size=n_build).astype(np.int32)
    pk = rng.integers(-25_000, 25_000, size=n_probe).astype(np.int32)
    join = qd.PartitionedHashJoin(capi, T.INT, 2 * n_build, group=group)
    join.build(torch.from_numpy(bk).to(dev), rank * n_build)
    pt, bt, op, ob, cnt = join.probe(torch.from_numpy(pk).to(dev), rank * n_probe)   # counts first: duplicates
    gp, gb = join.materialize(pt, bt, op, ob, cnt)
    save.update(h_build=bk, h_probe=pk, h_pp=gp.cpu().numpy(), h_pb=gb.cpu().numpy())

    # ---- BASELINE config 4: partitioned join with payload columns, strided directly addressed tables
    orders_per_rank = 40_000
    c4 = plans.generate_c4_inputs(dev, orders_per_rank, rank)
    pj = plans.PartitionedJoin(capi, orders_per_rank * world, orders_per_rank, group=group, dense=True)
    assert pj.join.table is not None
    cols, moved = pj.step(c4, rank * orders_per_rank, 0)
    assert plans.PartitionedJoin.check(cols)
    assert bool(((cols[0] % world) == rank).all())                # every output row sits on the rank that owns its key
    save.update(c4_o_key=c4["o_orderkey"].cpu().numpy(), c4_l_key=c4["l_orderkey"].cpu().numpy(),
                c4_l_pay=c4["l_payload"].cpu().numpy(), c4_out_key=cols[0].cpu().numpy(), c4_out_o=cols[1].cpu().numpy(),
                c4_out_l=cols[2].cpu().numpy(), c4_moved=np.int64(moved))

    # ---- broadcast join
    bj = qd.BroadcastHashJoin(capi, T.INT, n_build * world, group=group)
    assert bj.build(torch.from_numpy(bk).to(dev), rank * n_build) == n_build * world
    total_b = int(bj.table.probe_count(torch.from_numpy(pk).to(dev)).item())
    _, _, op3, ob3, cnt3 = bj.probe(torch.from_numpy(pk).to(dev), rank * n_probe, capacity=total_b)
    gp3, gb3 = bj.materialize(None, None, op3, ob3, cnt3)
    save.update(b_pp=gp3.cpu().numpy(), b_pb=gb3.cpu().numpy())

    # ---- Q1 state: partial aggregates of every rank merged by all-gather + import-merge (real AggState images)
    n = 200_000 + 1000 * rank
    k1 = rng.choice(np.frombuffer(b"ANR", dtype=np.uint8), size=n)
    k2 = rng.choice(np.frombuffer(b"FO", dtype=np.uint8), size=n)
    q1_cols = [k1, k2, rng.integers(1, 51, size=n).astype(np.float64), np.round(rng.uniform(900, 105000, size=n), 2),
               rng.integers(0, 11, size=n) / 100.0, rng.integers(0, 9, size=n) / 100.0]
    st = capi.AggState(q1_config())
    st.update([torch.from_numpy(c).to(dev) for c in q1_cols], n)
    qd.merge_agg_state_images(capi, st, group=group)
    keys, vals, _, groups = st.finalize(dev)
    g = int(groups.item())
    for i, c in enumerate(q1_cols):
        save[f"q1_col{i}"] = c
    for i, k in enumerate(keys):
        save[f"q1_key{i}"] = k.cpu().numpy()[:g]
    for i, v in enumerate(vals):
        save[f"q1_val{i}"] = v.cpu().numpy()[:g]

    # ---- dense (CollisionFreeVector) state: reduce-scatter, rank r finalizes key range r
    entries = 5_003
    dn = 120_000
    dkeys = rng.integers(0, entries, size=dn).astype(np.int32)
    dvals = rng.normal(size=dn)
    dcfg = T.make_agg_config(T.AGG_COLLISION_FREE, [(T.INT, None), (T.DOUBLE, None)], keys=[0],
                             aggs=[(T.AGG_COUNT_STAR, None), (T.AGG_SUM, T.col(1)), (T.AGG_MIN, T.col(1))], num_entries=entries)
    ds = capi.AggState(dcfg)
    ds.update([torch.from_numpy(dkeys).to(dev), torch.from_numpy(dvals).to(dev)], dn)
    qd.reduce_scatter_dense_state(ds, dev, group=group)
    dk, dv, _, dg = ds.finalize(dev, partition=rank, num_partitions=world)
    dg = int(dg.item())
    save.update(d_keys_in=dkeys, d_vals_in=dvals, d_key=dk[0].cpu().numpy()[:dg], d_cnt=dv[0].cpu().numpy()[:dg],
                d_sum=dv[1].cpu().numpy()[:dg], d_min=dv[2].cpu().numpy()[:dg])

    # ---- BASELINE config 5: Q3 with LIP filters, broadcast build sides, reduce-scatter of the dense partial aggregates
    q3_in = plans.generate_q3_inputs(dev, 0.02, rank, world)
    for fused in (True, False):
        q3 = plans.DistributedQ3(capi, q3_in["customers_total"], q3_in["orders_total"], group=group, use_lip=True, fused=fused)
        res = q3.run(q3_in, tid_base_orders=rank * q3_in["o_orderkey"].numel())
        tag = "f" if fused else "g"
        # this rank's groups after the merge (all of them, for the parent's comparison)
        keys_r, vals_r, _, groups_r = q3.state.finalize(dev, partition=rank, num_partitions=world)
        gr = int(groups_r.item())
        save.update({f"q3{tag}_keys": keys_r[0].cpu().numpy()[:gr], f"q3{tag}_rev": vals_r[0].cpu().numpy()[:gr],
                     f"q3{tag}_pairs": np.int64(res["pairs"]), f"q3{tag}_top_keys": res["top_keys"].cpu().numpy(),
                     f"q3{tag}_top_rev": res["top_revenue"].cpu().numpy()})
    for k, v in q3_in.items():
        if torch.is_tensor(v):
            save["q3in_" + k] = v.cpu().numpy()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **save)
    torch.cuda.synchronize()
    if group is not None:
        group.comm.close()
    dist.barrier()
    dist.destroy_process_group()
    print(f"RANKS_OK rank {rank}")


if __name__ == "__main__":
    main()
