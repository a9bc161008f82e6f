#!/usr/bin/env python3
"""bench.py — the headline measurement of BASELINE.json:
"probe+aggregate rows/s at TPC-H SF100 (Q1,Q3); % HBM roofline @1/2/4/8 GPU".

One step = one pass of the hot path over one batch of synthetic, HBM-resident
input, through the C ABI (quickstep_amd.capi -> libqsx.so):

  C2  hash join, Q3 customer⋈orders shape: clear + build a 1 M-key INT table,
      probe it with 100 M INT keys (match rate 1.0), emit (probe_tid, build_tid);
  C3  aggregation, Q1 lineitem shape: 600 M rows, GROUP BY two CHAR(1) keys,
      SUM(qty), SUM(price), SUM(price*(1-disc)), SUM(price*(1-disc)*(1+tax)),
      AVG(qty), AVG(price), AVG(disc), COUNT(*), then finalize.

value = (probe rows + aggregated rows) of all ranks / wall time of the step.
With --gpus N > 1 (launched by torch.distributed.run, one rank per GPU) every
rank holds the same per-GPU amount of rows (weak scaling); build and probe rows
are shuffled on the join key across ranks (K9 scatter + RCCL all-to-all) and
the partial Q1 states are merged across ranks (all-gather + import-merge).

The line also carries `roofline` for the dominant kernel (the aggregation
update kernel; algorithmic 34 B/row) with its duration measured
by HIP events on the launch stream, and `cpu_baseline`: the CPU oracle (a port
of the reference algorithms, oracle/) timed on this host's cores on a bounded
sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import quickstep_amd.capi as capi  # noqa: E402
from quickstep_amd import types as T  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling
Q1_BYTES_PER_ROW = 34          # 1 + 1 + 4 * 8


def q1_config():
    return T.make_agg_config(
        T.AGG_COMPACT_KEY,
        columns=[(T.CHAR, 1), (T.CHAR, 1), (T.DOUBLE, None), (T.DOUBLE, None), (T.DOUBLE, None), (T.DOUBLE, None)],
        keys=[0, 1],
        instrs=[(T.EX_SUB, 0, T.const(0), T.col(4)), (T.EX_MUL, 1, T.col(3), T.temp(0)),
                (T.EX_ADD, 2, T.const(0), T.col(5)), (T.EX_MUL, 3, T.temp(1), T.temp(2))],
        consts=[1.0],
        aggs=[(T.AGG_SUM, T.col(2)), (T.AGG_SUM, T.col(3)), (T.AGG_SUM, T.temp(1)), (T.AGG_SUM, T.temp(3)),
              (T.AGG_AVG, T.col(2)), (T.AGG_AVG, T.col(3)), (T.AGG_AVG, T.col(4)), (T.AGG_COUNT_STAR, None)],
        est_groups=6)


def gen_q1_columns_gpu(n, dev, seed):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    probs = torch.tensor([0.2466, 0.0065, 0.5005, 0.2464], device=dev)
    chunk = 50_000_000
    k1 = torch.empty(n, dtype=torch.uint8, device=dev)
    k2 = torch.empty(n, dtype=torch.uint8, device=dev)
    qty = torch.empty(n, dtype=torch.float64, device=dev)
    price = torch.empty(n, dtype=torch.float64, device=dev)
    disc = torch.empty(n, dtype=torch.float64, device=dev)
    tax = torch.empty(n, dtype=torch.float64, device=dev)
    m1 = torch.tensor(list(b"ANNR"), dtype=torch.uint8, device=dev)
    m2 = torch.tensor(list(b"FFOF"), dtype=torch.uint8, device=dev)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        combo = torch.multinomial(probs, e - s, replacement=True, generator=g)
        k1[s:e] = m1[combo]
        k2[s:e] = m2[combo]
        qty[s:e] = torch.randint(1, 51, (e - s,), device=dev, generator=g).double()
        price[s:e] = (torch.rand(e - s, device=dev, generator=g, dtype=torch.float64) * 104100 + 900).mul(100).round().div(100)
        disc[s:e] = torch.randint(0, 11, (e - s,), device=dev, generator=g).double() / 100
        tax[s:e] = torch.randint(0, 9, (e - s,), device=dev, generator=g).double() / 100
        del combo
    return [k1, k2, qty, price, disc, tax]


def gen_q1_columns_cpu(n, seed):
    rng = np.random.default_rng(seed)
    combo = rng.choice(4, size=n, p=[0.2466, 0.0065, 0.5005, 0.2464])
    return [np.frombuffer(b"ANNR", dtype=np.uint8)[combo], np.frombuffer(b"FFOF", dtype=np.uint8)[combo],
            rng.integers(1, 51, size=n).astype(np.float64), np.round(rng.uniform(900, 105000, size=n), 2),
            rng.integers(0, 11, size=n) / 100.0, rng.integers(0, 9, size=n) / 100.0]


def usable_cores():
    """Cores this process may really use: the affinity mask and the cgroup CPU quota bound os.cpu_count() (a box that
    shows 256 CPUs but grants a fraction of them runs 256 compute-bound threads at that fraction's speed)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            text = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = text[0], int(text[1])
            else:
                quota, period = text[0], int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1") and int(quota) > 0:
                n = min(n, max(1, int(quota) // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_baseline(args):
    """Oracle (port of the reference CPU algorithms) on a bounded sample, all usable host cores."""
    from oracle import pyoracle as O
    threads = usable_cores()
    rng = np.random.default_rng(3)
    build = rng.permutation(args.build_rows).astype(np.int32)
    block_join = 1_048_576           # 4 MB blocks of INT keys
    block_agg = 4 * 1024 * 1024 // Q1_BYTES_PER_ROW
    # calibrate on a small slice, then size the sample for ~8 s per operator
    probe_small = rng.integers(0, int(args.build_rows / args.match), size=2_000_000).astype(np.int32)
    r = O.bench_join(build, probe_small, block_join, threads)
    rate_p = probe_small.size / max(r["probe_seconds"], 1e-6)
    n_probe = int(min(args.probe_rows, max(4_000_000, rate_p * args.cpu_seconds)))
    probe = rng.integers(0, int(args.build_rows / args.match), size=n_probe).astype(np.int32)
    best = None
    for _ in range(2):
        r = O.bench_join(build, probe, block_join, threads)
        best = r if best is None or r["probe_seconds"] < best["probe_seconds"] else best
    rate_p = n_probe / best["probe_seconds"]
    rate_b = args.build_rows / best["build_seconds"]
    cfg = q1_config()
    cols_small = gen_q1_columns_cpu(2_000_000, 4)
    secs, st = O.bench_agg(cfg, cols_small, 2_000_000, block_agg, threads)
    st.close()
    rate_a = 2_000_000 / max(secs, 1e-6)
    n_agg = int(min(args.agg_rows, max(4_000_000, rate_a * args.cpu_seconds), 60_000_000))
    cols = gen_q1_columns_cpu(n_agg, 4)
    best_a = None
    for _ in range(2):
        secs, st = O.bench_agg(cfg, cols, n_agg, block_agg, threads)
        st.close()
        best_a = secs if best_a is None else min(best_a, secs)
    rate_a = n_agg / best_a
    mix = (args.probe_rows + args.agg_rows) / (args.probe_rows / rate_p + args.agg_rows / rate_a)
    return {
        "value": mix, "unit": "rows/s", "cores": threads, "kind": "port",
        "sample": f"oracle (CPU restatement of SimpleScalarSeparateChaining probe + ThreadPrivateCompactKey aggregation), "
                  f"{threads} worker threads, block-at-a-time; join {args.build_rows} x {n_probe} probe rows, "
                  f"aggregation {n_agg} rows; value = same probe:aggregate row mix as the GPU step",
        "probe_rows_per_s": rate_p, "build_rows_per_s": rate_b, "aggregate_rows_per_s": rate_a,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--build-rows", type=int, default=1_000_000)
    ap.add_argument("--probe-rows", type=int, default=100_000_000)
    ap.add_argument("--agg-rows", type=int, default=600_000_000)
    ap.add_argument("--match", type=float, default=1.0)
    ap.add_argument("--join-table", choices=["dense", "hashed"], default="dense",
                    help="dense: the build key (custkey) has exact min/max statistics -> directly addressed table "
                         "(qsx_join_table_create_dense); hashed: open-addressing table (qsx_join_table_create)")
    ap.add_argument("--join-plan", choices=["auto", "shuffle", "broadcast"], default="auto",
                    help="N > 1: shuffle = both sides repartitioned on the join key (K9 + RCCL all-to-all); broadcast = "
                         "all-gather of the build side, probe rows stay where they are (the reference's broadcast join, "
                         "BuildHashOperator.hpp:99,146-152); auto = broadcast while the gathered build side has <= 16 Mi rows")
    ap.add_argument("--other-plan-leg", action="store_true",
                    help="N > 1: after the timed region also run the join plan that was NOT picked, untimed, and report its cost")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if capi.device_count() < 1:
        raise SystemExit("libqsx.so sees no gfx950 device; there is no CPU path to benchmark")
    # QSX_BENCH_FORCE_DISTRIBUTED=1 runs the multi-GPU code path (RCCL shuffle + merge) even with one rank:
    # used to validate that path on the 1-GPU box (torch.distributed.run --nproc-per-node 1).
    distributed = world > 1 or os.environ.get("QSX_BENCH_FORCE_DISTRIBUTED") == "1"
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=dev)
        from quickstep_amd import distributed as qd

    # ---- synthetic inputs, resident in HBM before the timed region -----------
    g = torch.Generator(device=dev)
    g.manual_seed(2 + rank)
    key_space = args.build_rows * world
    build_keys = (torch.randperm(args.build_rows, device=dev, generator=g, dtype=torch.int32) * world + rank
                  if distributed else torch.randperm(args.build_rows, device=dev, generator=g, dtype=torch.int32))
    g.manual_seed(3 + rank)
    probe_keys = torch.randint(0, int(key_space / args.match), (args.probe_rows,), device=dev, generator=g,
                               dtype=torch.int32)
    agg_cols = gen_q1_columns_gpu(args.agg_rows, dev, 4 + rank)
    torch.cuda.synchronize()

    cfg = q1_config()
    state = capi.AggState(cfg)
    main_stream = torch.cuda.current_stream()
    dense = args.join_table == "dense"
    plan = args.join_plan
    if plan == "auto":
        plan = "broadcast" if args.build_rows * world <= 16 * 1024 * 1024 else "shuffle"
    # A second stream for the aggregation only pays next to the xGMI-bound shuffle.  Next to a local probe (broadcast
    # plan) the two kernels fight over L2: measured 6.5 ms per step side by side against 4.4 ms back to back on one GPU.
    agg_stream = torch.cuda.Stream(device=dev) if distributed and plan == "shuffle" else main_stream
    if distributed:
        def make_join(which):
            if which == "broadcast":
                return qd.BroadcastHashJoin(capi, T.INT, args.build_rows * world, key_domain=(0, key_space - 1) if dense else None)
            return qd.PartitionedHashJoin(capi, T.INT, 2 * args.build_rows, key_domain=(0, key_space - 1) if dense else None)
        join = make_join(plan)
        capacity = int(args.probe_rows * 1.25)
    else:
        table = capi.JoinTable(T.INT, args.build_rows, key_range=(0, args.build_rows - 1) if dense else None)
        capacity = args.probe_rows
        out = (torch.empty(capacity, dtype=torch.int32, device=dev), torch.empty(capacity, dtype=torch.int32, device=dev),
               torch.zeros(1, dtype=torch.int64, device=dev))

    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
    phase_ms = {"build": 0.0, "probe": 0.0, "aggregate_update": 0.0, "finalize": 0.0, "shuffle_build": 0.0,
                "shuffle_probe": 0.0, "merge": 0.0}
    results = {}
    recorded = []   # per timed step: (phase, start event, end event) on the stream the kernels were launched on

    def step(timed):
        if distributed:
            # shuffle plan: join on the main stream, aggregation on its own stream (the shuffle is xGMI-bound, the
            # aggregation HBM-bound, so the two overlap); broadcast plan: agg_stream IS the main stream.
            agg_stream.wait_stream(main_stream)
            with torch.cuda.stream(agg_stream):
                a0, a1, a2 = ev(), ev(), ev()
                a0.record()
                state.clear()
                state.update(agg_cols, args.agg_rows)
                a1.record()
                qd.merge_agg_state_images(capi, state)
                fin = state.finalize(dev, capacity=16)
                a2.record()
            e0, e1, e2 = ev(), ev(), ev()
            e0.record()
            nb = join.build(build_keys, rank * args.build_rows)
            e1.record()
            probe_tids, build_tids, op, ob, cnt = join.probe(probe_keys, rank * args.probe_rows, capacity=capacity)
            e2.record()
            main_stream.wait_stream(agg_stream)
            results.update(matches=cnt, groups=fin[3], built=nb)
            if timed:   # events are only READ after the timed loop (reading synchronises)
                recorded.append((("shuffle_build", e0, e1), ("shuffle_probe", e1, e2), ("aggregate_update", a0, a1),
                                 ("merge", a1, a2)))
            return
        e = [ev() for _ in range(6)]
        e[0].record()
        table.clear()
        table.build(build_keys)
        e[1].record()
        _, _, cnt = table.probe(probe_keys, capacity=capacity, out=out)
        e[2].record()
        state.clear()
        e[3].record()
        state.update(agg_cols, args.agg_rows)
        e[4].record()
        fin = state.finalize(dev, capacity=16)
        e[5].record()
        results.update(matches=cnt, groups=fin[3], fin=fin)
        if timed:
            recorded.append((("build", e[0], e[1]), ("probe", e[1], e[2]), ("aggregate_update", e[3], e[4]),
                             ("finalize", e[4], e[5])))

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)      # HIP events are recorded inside the timed region, read below
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    for per_step in recorded:
        for name, start, end in per_step:
            phase_ms[name] += start.elapsed_time(end)
    for k in phase_ms:
        phase_ms[k] /= args.steps

    other_plan_ms = None
    if distributed and world > 1 and args.other_plan_leg:
        # the plan the rule above did not pick, untimed leg (2 runs, second one measured): what the shuffle costs here.
        # Never allowed to take the headline measurement down with it.
        try:
            other = make_join("shuffle" if plan == "broadcast" else "broadcast")
            for it in range(2):
                torch.cuda.synchronize()
                dist.barrier()
                t1 = time.perf_counter()
                other.build(build_keys, rank * args.build_rows)
                other.probe(probe_keys, rank * args.probe_rows, capacity=capacity)
                torch.cuda.synchronize()
                other_plan_ms = (time.perf_counter() - t1) * 1e3
            del other
        except Exception as exc:  # noqa: BLE001
            other_plan_ms = f"failed: {exc!r}"[:200]
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        m = results["matches"].clone()
        dist.all_reduce(m, op=dist.ReduceOp.SUM)
        matches = int(m.item())
    else:
        matches = int(results["matches"].item())

    # ---- sanity: the timed work really produced the right shape of result ----
    expected_matches = None
    if args.match == 1.0:
        expected_matches = args.probe_rows * world
        assert matches == expected_matches, (matches, expected_matches)
    assert int(results["groups"].item()) == 4

    rows_per_step = (args.probe_rows + args.agg_rows) * world
    value = rows_per_step * args.steps / elapsed
    agg_s = phase_ms["aggregate_update"] / 1e3
    agg_gbs = Q1_BYTES_PER_ROW * args.agg_rows / agg_s / 1e9
    line = {
        "metric": "probe+aggregate rows/s at TPC-H SF100 (Q1,Q3)",
        "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "i32 keys / f64 sums", "data": "synthetic",
        "config": {
            "workload": f"C2 HashJoinOperator {args.build_rows} x {args.probe_rows} INTEGER inner equi-join (match rate "
                        f"{args.match}) + C3 AggregationOperator Q1 shape over {args.agg_rows} rows, per GPU",
            "build_rows": args.build_rows * world, "probe_rows": args.probe_rows * world,
            "aggregate_rows": args.agg_rows * world, "matches": matches,
            "parallelism": "1 GPU" if world == 1 else (
                f"{world} GPUs, " + ("broadcast join (all-gather of the build side, local probe)" if plan == "broadcast"
                                     else "join-key all-to-all shuffle of both sides") + " + partial-aggregate all-gather merge"),
        },
        "roofline": {
            "kernel": "agg_hash_shape_fixed_kernel<ShapeTpchQ1,4,16,4,1> (qsx_agg_update; AOT plan shape of the Q1 aggregation with its launch geometry as constants, same body as the interpreter kernel)", "bound": "hbm",
            "achieved": agg_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": agg_gbs / HBM_PEAK_GBS,
            "algorithmic_bytes_per_row": Q1_BYTES_PER_ROW, "rows_per_launch": args.agg_rows,
            "avg_launch_ms": phase_ms["aggregate_update"], "traffic": None,
        },
        "phases_ms": phase_ms,
    }
    if other_plan_ms is not None:
        line["join_plan"] = {"used": plan, "other_plan_build_plus_probe_ms": other_plan_ms}
    traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(traffic_file):
        try:
            tr = json.load(open(traffic_file))
            if tr.get("rows_per_launch") == args.agg_rows:
                line["roofline"]["traffic"] = tr.get("hbm_bytes_per_launch")
                line["roofline"]["traffic_source"] = tr.get("source")
        except Exception:
            pass
    if not distributed:
        probe_s = phase_ms["probe"] / 1e3
        probe_bytes = 4 * args.probe_rows + 8 * matches
        line["probe"] = {
            "rows_per_s": args.probe_rows / probe_s, "ms": phase_ms["probe"],
            "table": "directly addressed (exact min/max statistics of the build key)" if dense else "hashed",
            "roofline": {"kernel": "dense_probe_kernel<int,0> (qsx_join_probe)" if dense else
                         "probe_kernel<IntUnits,0> (qsx_join_probe)", "bound": "hbm",
                         "achieved": probe_bytes / probe_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": probe_bytes / probe_s / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_bytes": "4*N_probe + 8*N_match (hash-table traffic excluded)"},
        }
        if dense:
            # the same probe against the hashed table (what a build side without exact statistics gets), untimed leg
            hashed = capi.JoinTable(T.INT, args.build_rows)
            hashed.build(build_keys)
            hashed.probe(probe_keys, capacity=capacity, out=out)
            h0, h1 = ev(), ev()
            h0.record()
            for _ in range(3):
                hashed.probe(probe_keys, capacity=capacity, out=out)
            h1.record()
            torch.cuda.synchronize()
            line["probe"]["hashed_table_ms"] = h0.elapsed_time(h1) / 3
            hashed.close()
        line["build"] = {"rows_per_s": args.build_rows / (phase_ms["build"] / 1e3), "ms": phase_ms["build"]}
        line["aggregate"] = {"rows_per_s": args.agg_rows / agg_s, "ms": phase_ms["aggregate_update"],
                             "finalize_ms": phase_ms["finalize"]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(args)
    if rank == 0:
        print(json.dumps(line))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
