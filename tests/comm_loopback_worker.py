"""Worker of tests/test_gpu_comm.py::test_c_abi_collectives_at_world_2_and_3: one of 2 or 3 rank processes sharing cuda:0,
libqsx.so bound to the tests' loopback transport (QSX_RCCL_LIBRARY) — so that the N > 1 branches of the C ABI's
multi-GPU entry points run: qsx_exchange_counts, qsx_alltoallv (ragged, with empty pieces), qsx_allgather,
qsx_bitmap_allreduce_or, qsx_agg_reduce_scatter (key ranges that do not divide, range boundaries inside a 64-bit
existence word, fewer keys than ranks, MIN / MAX columns, states without them) and qsx_agg_allgather_merge (images of
different sizes).  Every result is compared, bit for bit, with what quickstep_amd/distributed.py produces for the same
inputs through torch.distributed (gloo, host staging) — the two routes must not drift — and saved for the parent, which
checks the union against numpy / the CPU oracle."""
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

DENSE_CASES = {          # name: (entries, rows per rank, aggregates over columns (key INT, a DOUBLE, b LONG))
    "ragged": (5_003, 60_000, [(T.AGG_COUNT_STAR, None), (T.AGG_SUM, T.col(1)), (T.AGG_MIN, T.col(1)), (T.AGG_MAX, T.col(2)),
                               (T.AGG_SUM, T.col(2))]),
    "sums_only": (130, 5_000, [(T.AGG_SUM, T.col(1)), (T.AGG_SUM, T.col(2))]),
    "fewer_keys_than_ranks": (2, 1_000, [(T.AGG_COUNT_STAR, None), (T.AGG_MIN, T.col(2)), (T.AGG_AVG, T.col(1))]),
    "one_word": (64, 3_000, [(T.AGG_MAX, T.col(1)), (T.AGG_COUNT_STAR, None)]),
}


def dense_config(entries, aggs):
    return T.make_agg_config(T.AGG_COLLISION_FREE, [(T.INT, None), (T.DOUBLE, None), (T.LONG, None)], keys=[0], aggs=aggs, num_entries=entries)


def dense_inputs(name, rank):
    entries, n, _ = DENSE_CASES[name]
    rng = np.random.default_rng(sum(name.encode()) + 31 * rank)
    keys = rng.integers(0, entries, size=n).astype(np.int32)
    if name == "ragged":
        keys[keys % 7 == rank] = 0          # holes that differ by rank: existence bits really come from different ranks
    a = rng.integers(-4096, 4096, size=n) / 64.0       # multiples of 1/64: sums are exact in any order
    b = rng.integers(-2**40, 2**40, size=n)
    return keys, a, b


def hash_config():
    return T.make_agg_config(T.AGG_GENERIC, [(T.INT, None), (T.DOUBLE, None), (T.LONG, None)], keys=[0],
                             aggs=[(T.AGG_SUM, T.col(1)), (T.AGG_COUNT_STAR, None), (T.AGG_MIN, T.col(2)), (T.AGG_MAX, T.col(1))], est_groups=64)


def hash_inputs(rank):
    rng = np.random.default_rng(900 + rank)
    groups = 40 if rank != 1 else 6_000        # rank 1's table outgrows its estimate: images of different sizes
    n = 50_000
    return rng.integers(-groups, groups, size=n).astype(np.int32), rng.integers(-4096, 4096, size=n) / 64.0, rng.integers(-2**40, 2**40, size=n)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out_dir = sys.argv[1]
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    assert os.environ.get("QSX_RCCL_LIBRARY")
    dist.init_process_group(backend="gloo")
    group = qd.CapiGroup.from_torch_group(capi, dev)
    comm = group.comm
    save = {}

    # ---- counts, all-to-all(v): rank r sends (r * 7 + p * 3) % 5 * 1000 + p rows to p — zero for some pairs
    rows = lambda src, dst: ((src * 7 + dst * 3) % 5) * 1000 + (dst if (src + dst) % 3 else 0)   # noqa: E731
    send_rows = [rows(rank, p) for p in range(world)]
    recv_rows = [rows(p, rank) for p in range(world)]
    got_counts = comm.exchange_counts(torch.tensor(send_rows, dtype=torch.int64, device=dev))
    assert got_counts.tolist() == recv_rows, (got_counts.tolist(), recv_rows)
    assert qd.exchange_counts(torch.tensor(send_rows, dtype=torch.int64, device=dev)).tolist() == recv_rows      # torch route
    for dtype in (torch.int32, torch.int64, torch.float64, torch.uint8):
        value = lambda src, dst, n: ((torch.arange(n, device=dev) * 13 + src * 101 + dst * 7) % 251).to(dtype)   # noqa: E731
        col = torch.cat([value(rank, p, send_rows[p]) for p in range(world)]) if sum(send_rows) else torch.empty(0, dtype=dtype, device=dev)
        got = comm.alltoallv(col, send_rows, recv_rows)
        want = torch.cat([value(p, rank, recv_rows[p]) for p in range(world)]) if sum(recv_rows) else col[:0]
        assert torch.equal(got, want), f"alltoallv {dtype}"
        twin = torch.empty_like(want)
        qd.xfer.all_to_all_single(twin, col, output_split_sizes=recv_rows, input_split_sizes=send_rows)
        assert torch.equal(got, twin)
        gathered = comm.allgather(col[:17].contiguous()) if col.numel() >= 17 else None
        if gathered is not None:
            pieces = [value(r, 0, rows(r, 0))[:17] for r in range(world)]
            if all(rows(r, 0) >= 17 for r in range(world)):
                assert torch.equal(gathered, torch.cat(pieces))
    mine = torch.full((5,), rank + 1, dtype=torch.int64, device=dev)
    assert comm.allgather(mine).tolist() == [r + 1 for r in range(world) for _ in range(5)]

    # ---- bit vectors
    g = torch.Generator(device=dev)
    g.manual_seed(77 + rank)
    words = torch.randint(-2**62, 2**62, (1_237,), device=dev, generator=g, dtype=torch.int64) & torch.randint(-2**62, 2**62, (1_237,), device=dev, generator=g, dtype=torch.int64)
    twin = words.clone()
    save["or_in"] = words.cpu().numpy()
    comm.bitmap_allreduce_or(words)
    qd._allreduce_or(twin)                      # gloo: ReduceOp.BOR
    assert torch.equal(words, twin)
    save["or_out"] = words.cpu().numpy()

    # ---- dense states
    for name, (entries, n, aggs) in DENSE_CASES.items():
        keys, a, b = dense_inputs(name, rank)
        cols = [torch.from_numpy(keys).to(dev), torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)]
        st, tw = capi.AggState(dense_config(entries, aggs)), capi.AggState(dense_config(entries, aggs))
        st.update(cols, n)
        tw.update(cols, n)
        qd.reduce_scatter_dense_state(st, dev, group=group)          # qsx_agg_reduce_scatter
        qd.reduce_scatter_dense_state(tw, dev)                        # torch.distributed + distributed.py's own range moves
        assert torch.equal(st.export(dev), tw.export(dev)), f"dense state {name}: the C ABI and distributed.py disagree"
        fk, fv, fn, fg = st.finalize(dev, partition=rank, num_partitions=world)
        k = int(fg.item())
        save[f"dense_{name}_key"] = fk[0].cpu().numpy()[:k]
        for i, v in enumerate(fv):
            save[f"dense_{name}_val{i}"] = v.cpu().numpy()[:k]
            save[f"dense_{name}_null{i}"] = fn[i].cpu().numpy()[:k]
        # nothing outside the owned range survives: finalizing ALL partitions of this rank's state gives the same groups
        ak, _, _, ag = st.finalize(dev)
        assert int(ag.item()) == k and torch.equal(ak[0][:k], fk[0][:k])

    # ---- hash states
    hk, ha, hb = hash_inputs(rank)
    cols = [torch.from_numpy(hk).to(dev), torch.from_numpy(ha).to(dev), torch.from_numpy(hb).to(dev)]
    st, tw = capi.AggState(hash_config()), capi.AggState(hash_config())
    st.update(cols, hk.size)
    tw.update(cols, hk.size)
    qd.merge_agg_state_images(capi, st, group=group)                  # qsx_agg_allgather_merge
    qd.merge_agg_state_images(capi, tw)
    results = []
    for s_ in (st, tw):
        fk, fv, _, fg = s_.finalize(dev)
        k = int(fg.item())
        order = torch.argsort(fk[0][:k])
        results.append([fk[0][:k][order]] + [v[:k][order] for v in fv])
    for x, y in zip(*results):
        assert torch.equal(x, y), "hash state: the C ABI and distributed.py disagree"
    save["hash_key"] = results[0][0].cpu().numpy()
    for i, v in enumerate(results[0][1:]):
        save[f"hash_val{i}"] = v.cpu().numpy()

    # ---- failure agreement (qsx_comm_agree): one rank contributes a failure, EVERY rank gets an error, and the communicator
    # is still in step afterwards — no rank is left inside a collective the failing one never entered
    comm.agree(0)                                                     # all fine: returns on every rank
    failing = world - 1
    try:
        comm.agree(T.ERR_OUT_OF_MEMORY if rank == failing else 0)
    except capi.QsxError as e:
        if rank == failing:
            assert e.status == T.ERR_OUT_OF_MEMORY, e
        else:
            assert e.status == T.ERR_COMM and f"rank {failing} failed" in str(e), e
    else:
        raise AssertionError("an agreed failure must raise on every rank")
    comm.agree(0)
    assert comm.allgather(mine).tolist() == [r + 1 for r in range(world) for _ in range(5)]
    comm.synchronize()

    # ---- the watchdog (qsx_comm_synchronize): a stream that does not drain before QSX_COMM_TIMEOUT_MS — what a collective
    # whose peer never arrives looks like — ends the wait with QSX_ERR_COMM and aborts the communicator
    import time
    os.environ["QSX_COMM_TIMEOUT_MS"] = "150"
    short = qd.CapiGroup.from_torch_group(capi, dev).comm
    os.environ.pop("QSX_COMM_TIMEOUT_MS")
    short.agree(0)
    torch.cuda._sleep(int(4e9))                                      # ~2 s of device time on the current stream
    t0 = time.perf_counter()
    try:
        short.synchronize()
    except capi.QsxError as e:
        assert e.status == T.ERR_COMM and "QSX_COMM_TIMEOUT_MS" in str(e), e
    else:
        raise AssertionError("the watchdog did not fire")
    assert time.perf_counter() - t0 < 1.5, "the watchdog waited for the stream instead of its deadline"
    try:
        short.agree(0)
    except capi.QsxError as e:
        assert e.status == T.ERR_COMM and "aborted" in str(e), e
    else:
        raise AssertionError("an aborted communicator must refuse further calls")
    short.close()
    torch.cuda.synchronize()
    dist.barrier()

    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **save)
    torch.cuda.synchronize()
    comm.close()
    dist.barrier()
    dist.destroy_process_group()
    print(f"RANKS_OK rank {rank}")


if __name__ == "__main__":
    main()
