#!/usr/bin/env python3
"""bench.py — the measurements of BASELINE.json:
"probe+aggregate rows/s at TPC-H SF100 (Q1,Q3); % HBM roofline @1/2/4/8 GPU".

--config headline (default): one step = one pass of the hot path over one batch of synthetic, HBM-resident input,
through the C ABI (quickstep_amd.capi -> libqsx.so):

  C2  hash join, Q3 customer⋈orders shape: clear + build a 1 M-key INT table, probe it with 100 M INT keys (match
      rate 1.0), emit (probe_tid, build_tid);
  C3  aggregation, Q1 lineitem shape: 600 M rows, GROUP BY two CHAR(1) keys, SUM(qty), SUM(price),
      SUM(price*(1-disc)), SUM(price*(1-disc)*(1+tax)), AVG(qty), AVG(price), AVG(disc), COUNT(*), then finalize.

value = (probe rows + aggregated rows) of all ranks / wall time of the step.  With --gpus N > 1 (launched by
torch.distributed.run, one rank per GPU) every rank holds the same per-GPU amount of rows (weak scaling); build and probe
rows are SHUFFLED on the join key across ranks (K9 scatter + RCCL all-to-all(v), the exchange north_star names; --join-plan
broadcast = all-gather of the build side instead) and the partial Q1 states are merged (all-gather + import-merge).

--config c4: BASELINE config 4, the partitioned join orders ⋈ lineitem with one 8-byte payload column per side
(quickstep_amd/plans.py PartitionedJoin): per rank 18.75 M orders and ~75 M lineitems (SF100 over 8 ranks), shuffle forced.
--config c5: BASELINE config 5, TPC-H Q3 with LIP filters, broadcast build sides, dense group-by and a reduce-scatter of
the partial aggregates (plans.DistributedQ3): SF 37.5 per rank (SF300 over 8 ranks).
Per-rank sizes are fixed, so one rank on the 1-GPU box runs the same code path (every collective over one rank).

After the timed region the results of the last step are CHECKED (pair validity, permutation of the probe rows, COUNT /
SUM values against independent torch reductions) — a wrong result fails the run instead of printing a number.

With one GPU the headline line also carries `secondary`: the other BASELINE configurations under the same clock — Q1 over
code stripes (the reference's own lineitem layout, benchmarks/tpch/create.sql:69-121), the C3 minimal variant, C4 and C5 at
their per-rank sizes — each with its time, roofline, result check and an oracle cpu_baseline on a bounded sample
(--no-secondary skips them).

The line carries `roofline` for the dominant kernel with its duration measured by HIP events on the launch stream, and
`cpu_baseline`: the CPU oracle (a port of the reference algorithms, oracle/) timed on this host's cores on a bounded sample
of the same workload — 5 trials, mean of trials 2-4 in run order (benchmarks/tpch/process.py:11-45).
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))


def self_launch():
    """`python bench.py --gpus N` with N > 1 and no rank environment: start the N ranks ourselves — a CHILD process running
    torch.distributed.run (one rank per GPU over RCCL), started before this process has imported torch or touched the GPU
    (never an exec: a GPU-initialised process must not replace its program on this pool) — relay what the ranks print (rank
    0's JSON line is the last line of stdout) and exit with the launcher's status.  Under torch.distributed.run (WORLD_SIZE
    set) this is a no-op."""
    n = 1
    argv = sys.argv[1:]
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1 or "WORLD_SIZE" in os.environ:
        return
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    env["QSX_BENCH_SELF_LAUNCHED"] = "1"
    # --standalone: the launcher binds its own rendezvous port (no bind-then-close race between benches started together)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={n}", os.path.abspath(__file__)] + argv
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    last_json = None
    for ln in child.stdout:
        if ln.startswith("{") and '"metric"' in ln:
            last_json = ln           # held back so that it is the LAST line this process prints
        else:
            sys.stdout.write(ln)
    rc = child.wait()
    sys.stdout.flush()
    if last_json is not None:
        sys.stdout.write(last_json)
        sys.stdout.flush()
    sys.exit(rc if rc != 0 or last_json is not None else 1)


if __name__ == "__main__":
    self_launch()

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

sys.path.insert(0, ROOT)

import quickstep_amd.capi as capi  # noqa: E402
from quickstep_amd import types as T  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling
XGMI_PEAK_GBS = 7 * 153.0      # per GPU: 7 links x ~153 GB/s
Q1_BYTES_PER_ROW = 34          # 1 + 1 + 4 * 8
METRIC = "probe+aggregate rows/s at TPC-H SF100 (Q1,Q3)"


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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def trials_2_to_4(run, trials=5):
    """5 trials, mean of trials 2-4 in run order (first = warm-up, last dropped)."""
    times = [run() for _ in range(trials)]
    return sum(times[1:4]) / 3.0, times


def cpu_baseline(args):
    """Oracle (port of the reference CPU algorithms) on a bounded sample, all usable host cores."""
    from oracle import pyoracle as O
    threads = usable_cores()
    rng = np.random.default_rng(3)
    build = rng.permutation(args.build_rows).astype(np.int32)
    block_join = 1_048_576           # 4 MB blocks of INT keys
    block_agg = 4 * 1024 * 1024 // Q1_BYTES_PER_ROW
    # calibrate on a small slice, then size the sample so that the five trials of an operator take ~cpu_seconds in all
    per_trial = args.cpu_seconds / 5.0
    probe_small = rng.integers(0, int(args.build_rows / args.match), size=2_000_000).astype(np.int32)
    r = O.bench_join(build, probe_small, block_join, threads)
    rate_p = probe_small.size / max(r["probe_seconds"], 1e-6)
    n_probe = int(min(args.probe_rows, max(4_000_000, rate_p * per_trial)))
    probe = rng.integers(0, int(args.build_rows / args.match), size=n_probe).astype(np.int32)
    builds = []

    def one_join():
        r = O.bench_join(build, probe, block_join, threads)
        builds.append(r["build_seconds"])
        return r["probe_seconds"]
    probe_s, probe_trials = trials_2_to_4(one_join)
    rate_p = n_probe / probe_s
    rate_b = args.build_rows / (sum(builds[1:4]) / 3.0)
    cfg = q1_config()
    cols_small = gen_q1_columns_cpu(2_000_000, 4)
    secs, st = O.bench_agg(cfg, cols_small, 2_000_000, block_agg, threads)
    st.close()
    rate_a = 2_000_000 / max(secs, 1e-6)
    n_agg = int(min(args.agg_rows, max(4_000_000, rate_a * per_trial), 60_000_000))
    cols = gen_q1_columns_cpu(n_agg, 4)

    def one_agg():
        secs, st = O.bench_agg(cfg, cols, n_agg, block_agg, threads)
        st.close()
        return secs
    agg_s, agg_trials = trials_2_to_4(one_agg)
    rate_a = n_agg / agg_s
    mix = (args.probe_rows + args.agg_rows) / (args.probe_rows / rate_p + args.agg_rows / rate_a)
    return {
        "value": mix, "unit": "rows/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
        "sample": f"oracle (CPU restatement of SimpleScalarSeparateChaining probe + ThreadPrivateCompactKey aggregation), "
                  f"{threads} worker threads, block-at-a-time; join {args.build_rows} x {n_probe} probe rows, "
                  f"aggregation {n_agg} rows; 5 trials each, mean of trials 2-4 in run order; value = same "
                  f"probe:aggregate row mix as the GPU step",
        "probe_rows_per_s": rate_p, "build_rows_per_s": rate_b, "aggregate_rows_per_s": rate_a,
        "probe_trials_s": probe_trials, "aggregate_trials_s": agg_trials,
    }


def kernel_source_digest():
    """sha256 over the device sources of the aggregation kernel: profiles/traffic.json is only quoted while it matches."""
    h = hashlib.sha256()
    for name in ("agg_hash_update.hpp", "agg_common.hpp", "agg_shapes.hpp", "agg_translate.hpp", "device_common.hpp", "aggregate.hip"):
        with open(os.path.join(ROOT, "quickstep_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


# ------------------------------------------------------------------------------------------------------------------------
class Ctx:
    pass


def ev():
    return torch.cuda.Event(enable_timing=True)


def timed_loop(ctx, step, args):
    """W untimed steps, then exactly K timed steps between barrier + synchronize on both sides; MAX over ranks."""
    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    if ctx.distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)      # HIP events are recorded inside the timed region, read afterwards
    torch.cuda.synchronize()
    if ctx.distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if ctx.distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=ctx.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def phase_means(recorded, steps):
    out = {}
    for per_step in recorded:
        for name, start, end in per_step:
            out[name] = out.get(name, 0.0) + start.elapsed_time(end)
    return {k: v / steps for k, v in out.items()}


def all_sum(ctx, value):
    if not ctx.distributed:
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=ctx.dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


# ------------------------------------------------------------------------------------------------------------------------
def check_pairs(probe_keys, build_keys, out_p, out_b, count, key_space, build_base=0, probe_base=0):
    """Exact, any size: (1) every pair satisfies the join condition, (2) the probe tids are exactly the probe rows whose
    key has a build row, each once (the build keys are unique)."""
    k = int(count)
    p, b = out_p[:k].long() - probe_base, out_b[:k].long() - build_base
    assert bool((build_keys[b] == probe_keys[p]).all()), "a pair violates the join condition"
    expect = torch.nonzero(probe_keys < key_space).flatten() if key_space is not None else None
    got = torch.sort(p).values
    if expect is not None:
        assert got.numel() == expect.numel() and bool((got == expect).all()), "probe tids are not the matching probe rows, once each"
    return k


def check_q1(agg_cols, fin, rows):
    """COUNT(*) per group exact against the number of rows with the group's key; SUM(qty) exact (integer-valued doubles: every
    partial sum is an integer below 2^53); SUM(price), SUM(price*(1-disc)) within 1e-6 of a torch f64 reduction."""
    keys, vals, _, groups = fin
    g = int(groups.item())
    k1, k2, qty, price, disc, _tax = agg_cols
    combo = k1.long() * 256 + k2.long()
    got_combo = keys[0][:g].long() * 256 + keys[1][:g].long()
    assert int(vals[7][:g].sum().item()) == rows, "COUNT(*) does not add up to the rows aggregated"
    assert torch.unique(got_combo).numel() == g, "a group appears twice"
    # per group: masked torch reductions (a handful of groups; index_add_ would be 1.8 G atomics on four addresses)
    for i in range(g):
        mask = combo == got_combo[i]
        zero = torch.zeros((), dtype=torch.float64, device=qty.device)
        # COUNT(*) exact: with the counts adding up to the row count (above) no group is missing either
        assert vals[7][i].item() == int(mask.sum().item()), "COUNT(*) of a group differs from the rows with its key"
        assert vals[0][i].item() == torch.where(mask, qty, zero).sum().item(), "SUM(l_quantity) is not exact"
        for a, ref in ((1, torch.where(mask, price, zero).sum().item()),
                       (2, torch.where(mask, price * (1.0 - disc), zero).sum().item())):
            rel = abs(vals[a][i].item() - ref) / abs(ref)
            assert rel <= 1e-6, f"aggregate {a} of group {i}: relative error {rel}"
        del mask
    return g


# ------------------------------------------------------------------------------------------------------------------------
def run_headline(ctx, args):
    dev, rank, world, distributed = ctx.dev, ctx.rank, ctx.world, ctx.distributed
    if distributed:
        from quickstep_amd import distributed as qd
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
        plan = "shuffle"       # the exchange BASELINE.json names; broadcast is the cheaper plan for a 1 M-row build side
    # A second stream for the aggregation only pays next to the xGMI-bound shuffle.  Next to a local probe (broadcast
    # plan) the two kernels fight over L2: measured 6.5 ms per step side by side against 4.4 ms back to back on one GPU.
    agg_stream = torch.cuda.Stream(device=dev) if distributed and plan == "shuffle" else main_stream
    if distributed:
        def make_join(which):
            if which == "broadcast":
                return qd.BroadcastHashJoin(capi, T.INT, args.build_rows * world, group=ctx.group, key_domain=(0, key_space - 1) if dense else None)
            return qd.PartitionedHashJoin(capi, T.INT, 2 * args.build_rows, group=ctx.group, key_domain=(0, key_space - 1) if dense else None)
        join = make_join(plan)
        capacity = int(args.probe_rows * 1.25)
    else:
        table = capi.JoinTable(T.INT, args.build_rows, key_range=(0, args.build_rows - 1) if dense else None)
        capacity = args.probe_rows
        out = (torch.empty(capacity, dtype=torch.int32, device=dev), torch.empty(capacity, dtype=torch.int32, device=dev),
               torch.zeros(1, dtype=torch.int64, device=dev))

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
                qd.merge_agg_state_images(capi, state, group=ctx.group)
                fin = state.finalize(dev, capacity=16)
                a2.record()
            e0, e1, e2 = ev(), ev(), ev()
            e0.record()
            nb = join.build(build_keys, rank * args.build_rows)
            e1.record()
            probe_tids, build_tids, op, ob, cnt = join.probe(probe_keys, rank * args.probe_rows, capacity=capacity)
            e2.record()
            main_stream.wait_stream(agg_stream)
            results.update(matches=cnt, groups=fin[3], fin=fin, built=nb, pairs=(probe_tids, build_tids, op, ob))
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

    elapsed = timed_loop(ctx, step, args)
    phase_ms = phase_means(recorded, args.steps)

    other_plan_ms = None
    if distributed and world > 1 and args.other_plan_leg:
        # the plan that was not timed, untimed leg (2 runs, second one measured).  Never allowed to take the headline down.
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
    matches = all_sum(ctx, results["matches"].item())

    # ---- the timed work produced the right RESULT (last step), not just the right shape ----------------------------------
    checks = {}
    if not args.no_check:
        if not distributed:
            checks["pairs"] = check_pairs(probe_keys, build_keys, out[0], out[1], results["matches"].item(), key_space)
        else:
            # shuffled join: the pairs index the rows that arrived on this rank — every pair must satisfy the join condition
            # on the keys that arrived, and sit on the rank that owns its key; the pairs of all ranks together are one per
            # probe row that has a build row
            probe_tids, build_tids, op, ob = results["pairs"]
            k = int(results["matches"].item())
            if plan == "shuffle":
                pk, bk = join.probe_keys[op[:k].long()], join.build_keys[ob[:k].long()]
                assert bool((pk == bk).all()), "a pair violates the join condition"
                assert world & (world - 1) or bool(((pk & (world - 1)) == rank).all()), "a pair sits on the wrong rank"
            expected = all_sum(ctx, int((probe_keys < key_space).sum().item()))
            assert matches == expected, (matches, expected)
            checks["pairs"] = matches
        if args.match == 1.0:
            assert matches == args.probe_rows * world, (matches, args.probe_rows * world)
        if not distributed:
            checks["q1_groups"] = check_q1(agg_cols, results["fin"], args.agg_rows)
        else:
            fin = results["fin"]
            g_ = int(fin[3].item())
            assert all_sum(ctx, args.agg_rows) == int(fin[1][7][:g_].sum().item()), "merged COUNT(*) != rows of all ranks"
            checks["q1_groups"] = g_
    assert int(results["groups"].item()) == 4

    rows_per_step = (args.probe_rows + args.agg_rows) * world
    value = rows_per_step * args.steps / elapsed
    agg_s = phase_ms["aggregate_update"] / 1e3
    agg_gbs = Q1_BYTES_PER_ROW * args.agg_rows / agg_s / 1e9
    line = {
        "metric": METRIC,
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
            "kernel": "agg_hash_shape_fixed_kernel<ShapeTpchQ1,4,16,4,1,false> (qsx_agg_update; AOT plan shape of the Q1 aggregation with its launch geometry as constants, same body as the interpreter kernel)", "bound": "hbm",
            "achieved": agg_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": agg_gbs / HBM_PEAK_GBS,
            "algorithmic_bytes_per_row": Q1_BYTES_PER_ROW, "rows_per_launch": args.agg_rows,
            "avg_launch_ms": phase_ms["aggregate_update"], "traffic": None,
        },
        "phases_ms": phase_ms,
        "checked": checks,
    }
    line.setdefault("_probe_traffic", None)
    if other_plan_ms is not None:
        line["join_plan"] = {"used": plan, "other_plan_build_plus_probe_ms": other_plan_ms}
    if distributed and plan == "shuffle":
        moved = join.shuffled_bytes * (world - 1) / max(world, 1)
        sh_s = (phase_ms["shuffle_build"] + phase_ms["shuffle_probe"]) / 1e3
        line["alltoall"] = {"bytes_per_rank_per_step": moved, "GBps_per_rank": moved / sh_s / 1e9, "GBps_per_rank_over_shuffle_phases": moved / sh_s / 1e9,
                            "peak_GBps_per_rank": XGMI_PEAK_GBS,
                            "note": "phases include K9 scatter, counts exchange, build / probe kernels"}
    traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(traffic_file):
        try:
            tr = json.load(open(traffic_file))
            if tr.get("rows_per_launch") == args.agg_rows:
                if tr.get("kernel_source_digest") in (None, kernel_source_digest()):
                    line["roofline"]["traffic"] = tr.get("hbm_bytes_per_launch")
                    line["roofline"]["traffic_source"] = tr.get("source")
                    line["_probe_traffic"] = tr.get("probe")
                else:
                    line["roofline"]["traffic_source"] = "profiles/traffic.json is for other kernel sources: stale, not quoted"
        except Exception:
            pass
    if not distributed:
        probe_s = phase_ms["probe"] / 1e3
        probe_bytes = 4 * args.probe_rows + 8 * matches
        line["probe"] = {
            "rows_per_s": args.probe_rows / probe_s, "ms": phase_ms["probe"],
            "table": "directly addressed (exact min/max statistics of the build key)" if dense else "hashed",
            "roofline": {"kernel": "dense_probe_kernel<int,0> (qsx_join_probe)" if dense else
                         "probe_fp_kernel<0> (qsx_join_probe; bucketed table + fingerprint plane)", "bound": "hbm",
                         "achieved": probe_bytes / probe_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": probe_bytes / probe_s / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_bytes": "4*N_probe + 8*N_match (hash-table traffic excluded)"},
        }
        probe_traffic = line.pop("_probe_traffic", None)
        if probe_traffic is not None and dense and args.probe_rows == 100_000_000 and args.build_rows == 1_000_000:
            line["probe"]["roofline"]["traffic"] = probe_traffic.get("hbm_bytes_per_launch")
            line["probe"]["roofline"]["traffic_detail"] = probe_traffic
        if not args.no_probe_variants:
            line["probe"]["variants"] = probe_variants(ctx, args, build_keys, probe_keys, out)
        line["build"] = {"rows_per_s": args.build_rows / (phase_ms["build"] / 1e3), "ms": phase_ms["build"]}
        line["aggregate"] = {"rows_per_s": args.agg_rows / agg_s, "ms": phase_ms["aggregate_update"],
                             "finalize_ms": phase_ms["finalize"]}
    line.pop("_probe_traffic", None)
    if getattr(ctx, "keep_headline", False):
        ctx.headline = {"agg_cols": agg_cols, "fin": results["fin"]}
    return line


def probe_variants(ctx, args, build_keys, probe_keys, out):
    """The other §8(d) C2 legs, each timed with HIP events over 3 launches after one warm-up: hashed tables (a build side
    without statistics: over the dense keys, with and without the directly addressed shadow, and over sparse keys), match
    rate 0.2 (Q3's c_mktsegment filter), and the materialised join (+ one 8-byte payload gathered from each side, +16 B per
    match).  Every leg's pairs are checked against the join condition and the expected count."""
    dev = ctx.dev
    n = args.probe_rows
    res = {}

    def timed(fn, reps=3):
        fn()
        a, b = ev(), ev()
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    g = torch.Generator(device=dev)
    g.manual_seed(33)
    keys_02 = torch.randint(0, args.build_rows * 5, (n,), device=dev, generator=g, dtype=torch.int32)
    spread = lambda k: (k.long() * 2039 % (2**31 - 1)).to(torch.int32)   # noqa: E731  (a bijection: unique, sparse keys)
    legs = (("dense", True, "1", False),              # exact statistics from the optimizer
            ("hashed", False, "1", False),            # no statistics: the first probe gives the table a directly addressed shadow
            ("hashed_no_shadow", False, "0", False),  # QSX_JOIN_ADAPTIVE=0: the hashed kernels themselves on the same keys
            ("hashed_sparse_keys", False, "1", True))  # keys spread over the INT range: no shadow possible
    for name, dense, adaptive, sparse in legs:
        os.environ["QSX_JOIN_ADAPTIVE"] = adaptive
        bk = spread(build_keys) if sparse else build_keys
        t = capi.JoinTable(T.INT, args.build_rows, key_range=(0, args.build_rows - 1) if dense else None)
        t.build(bk)
        for m, keys in ((1.0, probe_keys), (0.2, keys_02)):
            if sparse:
                keys = spread(keys)
            ms = timed(lambda: t.probe(keys, capacity=n, out=out))
            matches = int(out[2].item())
            k = int(out[2].item())
            assert bool((bk[out[1][:k].long()] == keys[out[0][:k].long()]).all()), "a pair violates the join condition"
            assert matches == int((keys_02 < args.build_rows).sum().item()) if m == 0.2 else matches == n
            byts = 4 * n + 8 * matches
            res[f"{name}_m{m}"] = {"ms": ms, "matches": matches, "rows_per_s": n / ms * 1e3, "GBps": byts / ms / 1e6,
                                   "frac_of_hbm_peak": byts / ms / 1e6 / HBM_PEAK_GBS}
        if not dense:
            # clear + build + first probe: what the shadow costs a query that builds the table once
            def cycle():
                t.clear()
                t.build(bk)
                t.probe(probe_keys if not sparse else spread(probe_keys), capacity=n, out=out)
            res[f"{name}_clear_build_probe_ms"] = timed(cycle)
        if dense:
            # materialised: one 8-byte payload column from each side, gathered by the pair list (K5)
            pay_b = build_keys.long() * 3 + 1
            pay_p = probe_keys.long() * 7 + 2
            t.probe(probe_keys, capacity=n, out=out)
            k = int(out[2].item())
            ob = torch.empty(k, dtype=torch.int64, device=dev)
            op = torch.empty(k, dtype=torch.int64, device=dev)

            def materialise():
                t.probe(probe_keys, capacity=n, out=out)
                capi.gather(pay_b, out[1][:k], out=ob)
                capi.gather(pay_p, out[0][:k], out=op)
            ms = timed(materialise)
            assert bool((ob == probe_keys[out[0][:k].long()].long() * 3 + 1).all()) and bool((op == probe_keys[out[0][:k].long()].long() * 7 + 2).all())
            byts = 4 * n + 8 * k + 16 * k
            res["dense_m1.0_materialised"] = {"ms": ms, "matches": k, "GBps": byts / ms / 1e6,
                                              "frac_of_hbm_peak": byts / ms / 1e6 / HBM_PEAK_GBS}
            # the same output relation written by the probe itself (qsx_join_probe_project_blocks), and the operators' bench's
            # output relation (one INT attribute from each side)
            def projected():
                return t.probe_project_blocks([probe_keys], [[pay_p]], [[pay_b]], capacity=n)
            ms = timed(projected)
            (gp, gb), cnt = projected()
            assert int(cnt.item()) == k and bool((gb == (gp - 2) // 7 * 3 + 1).all()) and int(gp.sum().item()) == int(pay_p.sum().item())
            res["dense_m1.0_projected"] = {"ms": ms, "matches": k, "GBps": byts / ms / 1e6, "frac_of_hbm_peak": byts / ms / 1e6 / HBM_PEAK_GBS}
            ms = timed(lambda: t.probe_project_blocks([probe_keys], [[probe_keys]], [[build_keys]], capacity=n))
            res["dense_m1.0_projected_int_attributes"] = {"ms": ms, "matches": k, "GBps": (4 * n + 8 * k) / ms / 1e6}
            del pay_b, pay_p, ob, op, gp, gb
        t.close()
    os.environ.pop("QSX_JOIN_ADAPTIVE", None)
    return res


# ------------------------------------------------------------------------------------------------------------------------
def run_c4(ctx, args):
    from quickstep_amd import plans
    dev, rank, world = ctx.dev, ctx.rank, ctx.world
    n_o = args.c4_orders_per_rank
    inputs = plans.generate_c4_inputs(dev, n_o, rank)
    n_l = inputs["l_orderkey"].numel()
    torch.cuda.synchronize()
    pj = plans.PartitionedJoin(capi, n_o * world, n_o, group=ctx.group, dense=args.join_table == "dense", fused=not args.c4_unfused,
                               overlap=not args.c4_no_overlap)
    recorded, results = [], {}

    def step(timed):
        e0, e1 = ev(), ev()
        e0.record()
        cols, moved = pj.step(inputs, rank * n_o, 0)
        e1.record()
        results.update(cols=cols, moved=moved)
        if timed:
            recorded.append((("partitioned_join", e0, e1),))

    elapsed = timed_loop(ctx, step, args)
    phase_ms = phase_means(recorded, args.steps)
    # one more (untimed) step with HIP events around the device phases of the plan
    from quickstep_amd import distributed as qd
    qd.phases.enabled = True
    step(False)
    phase_ms.update({"  " + k: v for k, v in qd.phases.read().items()})
    qd.phases.enabled = False
    cols = results["cols"]
    out_rows = all_sum(ctx, cols[0].numel())
    total_lines = all_sum(ctx, n_l)
    if not args.no_check:
        assert plans.PartitionedJoin.check(cols), "a joined row violates the join condition"
        assert bool(((cols[0] & (world - 1)) == rank).all()) or world & (world - 1)
        assert out_rows == total_lines, (out_rows, total_lines)          # every lineitem row has exactly one order
    rows = (n_o * world + total_lines)
    moved = results["moved"] * (world - 1) / world                       # bytes this rank really sent to peers
    # HBM algorithmic bytes per rank: scan both relations once (4 + 8 B/row), write the join output (4 + 8 + 8 B/row)
    algo = (4 + 8) * (n_o + n_l) + (4 + 8 + 8) * cols[0].numel()
    step_s = elapsed / args.steps
    return {
        "metric": METRIC, "value": rows * args.steps / elapsed, "unit": "rows/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "i32 keys / 8-byte payloads", "data": "synthetic",
        "config": {"workload": f"C4 partitioned hash join orders ⋈ lineitem, {n_o} orders + {n_l} lineitems per GPU, one 8-byte "
                               "payload column per side, both sides shuffled on orderkey (K9 scatter + all-to-all(v))",
                   "orders": n_o * world, "lineitems": total_lines, "output_rows": out_rows,
                   "parallelism": f"{world} GPU(s), hash partition = orderkey & (P-1)"},
        "roofline": {"kernel": "whole step (K9 partition_scatter x2, build, probe, K5 gathers)", "bound": "hbm",
                     "achieved": algo / step_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": algo / step_s / 1e9 / HBM_PEAK_GBS,
                     "algorithmic_bytes_per_step": algo, "traffic": None},
        "checked": not args.no_check,
        "alltoall": {"bytes_sent_per_rank_per_step": moved, "GBps_per_rank": moved / step_s / 1e9, "GBps_per_rank_over_step": moved / step_s / 1e9,
                     "peak_GBps_per_rank": XGMI_PEAK_GBS},
        "phases_ms": phase_ms,
    }


def run_c5(ctx, args):
    from quickstep_amd import plans
    dev, rank, world = ctx.dev, ctx.rank, ctx.world
    inputs = plans.generate_q3_inputs(dev, args.c5_sf_per_rank, rank, world)
    torch.cuda.synchronize()
    os.environ.setdefault("QSX_AGG_JIT_MIN_ROWS", "0")
    q3 = plans.DistributedQ3(capi, inputs["customers_total"], inputs["orders_total"], group=ctx.group, use_lip=not args.no_lip, fused=True)
    recorded, results = [], {}

    def step(timed):
        e0, e1 = ev(), ev()
        e0.record()
        res = q3.run(inputs, tid_base_orders=rank * inputs["o_orderkey"].numel())
        e1.record()
        results.update(res)
        if timed:
            recorded.append((("q3", e0, e1),))

    elapsed = timed_loop(ctx, step, args)
    phase_ms = phase_means(recorded, args.steps)
    from quickstep_amd import distributed as qd
    qd.phases.enabled = True          # one more (untimed) step with HIP events around the phases of the plan
    step(False)
    phase_ms.update({"  " + k: v for k, v in qd.phases.read().items()})
    qd.phases.enabled = False
    rows_rank = inputs["c_custkey"].numel() + inputs["o_orderkey"].numel() + inputs["l_orderkey"].numel()
    rows = all_sum(ctx, rows_rank)
    pairs = all_sum(ctx, results["pairs"])
    groups = all_sum(ctx, results["groups"])
    if not args.no_check:
        # independent torch evaluation of the query on this rank's lineitems against the GLOBAL qualifying-order set
        c_ok = torch.zeros(inputs["customers_total"] + 1, dtype=torch.bool, device=dev)
        mine = inputs["c_custkey"][inputs["c_mktsegment"] == plans.SEG_BUILDING].long()
        c_ok[mine] = True
        if ctx.distributed:
            t = c_ok.to(torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            c_ok = t.bool()
        o_ok = torch.zeros(inputs["orders_total"] + 1, dtype=torch.bool, device=dev)
        sel = (inputs["o_orderdate"] < plans.DATE_CUT) & c_ok[inputs["o_custkey"].long()]
        o_ok[inputs["o_orderkey"][sel].long()] = True
        if ctx.distributed:
            t = o_ok.to(torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            o_ok = t.bool()
        l_sel = (inputs["l_shipdate"] > plans.DATE_CUT) & o_ok[inputs["l_orderkey"].long()]
        assert all_sum(ctx, int(l_sel.sum().item())) == pairs, "joined pairs differ from the torch evaluation"
        revenue = torch.zeros(inputs["orders_total"] + 1, dtype=torch.float64, device=dev)
        revenue.index_add_(0, inputs["l_orderkey"][l_sel].long(), (inputs["l_extendedprice"] * (1.0 - inputs["l_discount"]))[l_sel])
        if ctx.distributed:
            dist.all_reduce(revenue, op=dist.ReduceOp.SUM)
        assert int((revenue != 0).sum().item()) == groups, "group count differs from the torch evaluation"
        top = torch.topk(revenue, 10).values
        rel = ((results["top_revenue"] - top).abs() / top).max().item()
        assert rel <= 1e-6, f"top-10 revenue differs from the torch evaluation: {rel}"
        del revenue, c_ok, o_ok
    algo = plans.q3_input_bytes(inputs)
    step_s = elapsed / args.steps
    return {
        "metric": METRIC, "value": rows * args.steps / elapsed, "unit": "rows/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "i32 keys / f64 sums", "data": "synthetic",
        "config": {"workload": f"C5 TPC-H Q3 pipeline (3-way join + LIP filters + group-by l_orderkey + top 10) at SF "
                               f"{args.c5_sf_per_rank} per GPU, input rows = customer + orders + lineitem",
                   "input_rows": rows, "joined_pairs": pairs, "groups": groups,
                   "parallelism": f"{world} GPU(s): broadcast build sides (all-gather), LIP bit vectors OR-ed across ranks, dense "
                                  "partial aggregates merged by reduce-scatter, every rank finalizes its key range"},
        "roofline": {"kernel": "whole query (K1 selects, LIP, K3/K4 joins, K7 dense aggregation through the pair list, K10, top-k)",
                     "bound": "hbm", "achieved": algo / step_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": algo / step_s / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_step": algo, "traffic": None},
        "checked": not args.no_check,
        "collectives": {"bytes_per_rank_per_step": q3.comm_bytes, "GBps_per_rank_over_step": q3.comm_bytes / step_s / 1e9,
                        "peak_GBps_per_rank": XGMI_PEAK_GBS},
        "phases_ms": phase_ms,
    }


# ------------------------------------------------------------------------------------------------------------------------
# secondary: the other BASELINE configurations under the driver's clock (N = 1, headline config)
def launches_ms(fn, reps=5, warm=2):
    """Average duration of one launch of `fn` by HIP events on the launch stream (capi launches on torch's current stream)."""
    for _ in range(warm):
        fn()
    total = 0.0
    for _ in range(reps):
        a, b = ev(), ev()
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        total += a.elapsed_time(b)
    return total / reps


def hbm_roofline(kernel, bytes_per_launch, ms, **extra):
    gbs = bytes_per_launch / ms / 1e6
    out = {"kernel": kernel, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
           "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": ms}
    out.update(extra)
    return out


def sized_sample(run_small, small_units, seconds_per_trial, lo, hi):
    """Rows (orders, scale factor ...) for one trial of about `seconds_per_trial`, from one calibration run."""
    secs = max(run_small(), 1e-6)
    return int(min(hi, max(lo, small_units / secs * seconds_per_trial)))


def q1_coded_config():
    cfg = q1_config()
    for c in (2, 4, 5):                      # l_quantity, l_discount, l_tax: 1-byte dictionary codes (create.sql:69-121)
        cfg.column_code_width[c] = 1
    return cfg


def minimal_config():
    return T.make_agg_config(T.AGG_COMPACT_KEY, [(T.INT, None), (T.INT, None), (T.DOUBLE, None)], keys=[0, 1],
                             aggs=[(T.AGG_SUM, T.col(2)), (T.AGG_COUNT_STAR, None), (T.AGG_AVG, T.col(2))], est_groups=9)


def secondary_q1_coded(ctx, args, threads):
    """C3 over lineitem as the reference's DDL stores it: CompressedColumnStore, l_quantity / l_discount / l_tax as 1-byte
    dictionary codes, the keys CHAR(1), l_extendedprice a plain DOUBLE stripe: 1 + 1 + 1 + 8 + 1 + 1 = 13 B/row.  The
    columns are the headline's (codes of the same values), so the result must equal the headline's."""
    dev, n = ctx.dev, args.agg_rows
    k1, k2, qty, price, disc, tax = ctx.headline["agg_cols"]
    qty_c = (qty - 1.0).to(torch.uint8)
    disc_c = (disc * 100.0).round().to(torch.uint8)
    tax_c = (tax * 100.0).round().to(torch.uint8)
    dicts = [None, None, torch.arange(1, 51, device=dev, dtype=torch.float64), None,
             torch.arange(0, 11, device=dev, dtype=torch.float64) / 100, torch.arange(0, 9, device=dev, dtype=torch.float64) / 100]
    cols = [k1, k2, qty_c, price, disc_c, tax_c]
    jit_env = os.environ.get("QSX_AGG_JIT_MIN_ROWS")
    os.environ["QSX_AGG_JIT_MIN_ROWS"] = "0"      # the plan shape is built in the calling thread before the first launch
    st = capi.AggState(q1_coded_config())
    st.clear()
    st.update_coded(cols, dicts, n)               # first use: the shape's compile (or its load from QSX_JIT_CACHE_DIR)
    torch.cuda.synchronize()
    if jit_env is None:
        os.environ.pop("QSX_AGG_JIT_MIN_ROWS")
    else:
        os.environ["QSX_AGG_JIT_MIN_ROWS"] = jit_env

    def one():
        st.clear()
        st.update_coded(cols, dicts, n)
    ms_clear_update = launches_ms(one)
    ms = launches_ms(lambda: st.update_coded(cols, dicts, n), warm=0)
    st.clear()
    st.update_coded(cols, dicts, n)
    ck, cv, _, cg = st.finalize(dev, capacity=16)
    checked = False
    if not args.no_check:
        hk, hv, _, hg = ctx.headline["fin"]
        g = int(hg.item())
        assert int(cg.item()) == g, "groups over code stripes differ from the plain columns'"
        oc = torch.argsort(ck[0][:g].long() * 256 + ck[1][:g].long())
        oh = torch.argsort(hk[0][:g].long() * 256 + hk[1][:g].long())
        for a in range(len(hv)):
            x, y = cv[a][:g][oc], hv[a][:g][oh]
            if a in (0, 7):      # SUM(l_quantity) (integer-valued doubles) and COUNT(*): exact
                assert bool((x == y).all()), f"aggregate {a} over code stripes is not exact"
            else:
                assert bool(torch.allclose(x.double(), y.double(), rtol=1e-9, atol=0.0)), f"aggregate {a} over code stripes differs"
        checked = True
    st.close()
    blocks = q1_coded_as_blocks(ctx, args, cols, dicts, (ck, cv, cg))
    out = {"workload": f"C3 Q1 aggregation over CompressedColumnStore lineitem: {n} rows, 13 B/row (l_quantity / l_discount / "
                       "l_tax 1-byte dictionary codes decoded in the kernel)",
           "ms": ms, "ms_with_clear": ms_clear_update, "rows_per_s": n / ms * 1e3, "blocks": blocks,
           "roofline": hbm_roofline("agg_factored_direct_kernel<false,1,2,2,1,true> (qsx_agg_update_coded_sized: the aggregates factored through the "
                                    "dictionary codes, csrc/agg_factored.hpp; the call also launches factored_coef_kernel, ~10 us)",
                                    13 * n, ms, algorithmic_bytes_per_row=13),
           "checked": checked}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_q1_coded(args, threads)
    return out


def q1_coded_as_blocks(ctx, args, cols, dicts, single):
    """The same rows as the operators hand them over: a run of 4 MiB blocks (322 640 rows of 13 B), every block with dictionaries of
    its own (the reference compresses block by block, storage/CompressedBlockBuilder.cpp:300-368) — one
    qsx_agg_update_coded_blocks_sized call; the result must equal the single stripe's."""
    import ctypes as C
    dev, n = ctx.dev, args.agg_rows
    rows_per_block = 322_640
    starts = list(range(0, n, rows_per_block))
    nb, ncols = len(starts), len(cols)
    own = [[None if d is None else d.clone() for d in dicts] for _ in starts]
    # (the argument arrays are built once: filling them is Python's time, not the call's)
    a_rows = (C.c_int64 * nb)(*[min(n, lo + rows_per_block) - lo for lo in starts])
    a_cols, a_dicts, a_entries = (C.c_void_p * (nb * ncols))(), (C.c_void_p * (nb * ncols))(), (C.c_int32 * (nb * ncols))()
    for b, lo in enumerate(starts):
        for c in range(ncols):
            a_cols[b * ncols + c] = cols[c].data_ptr() + lo * cols[c].element_size()
            d = own[b][c]
            a_dicts[b * ncols + c] = d.data_ptr() if d is not None else None
            a_entries[b * ncols + c] = d.numel() if d is not None else 0
    st = capi.AggState(q1_coded_config())

    def update():
        rc = capi.lib.qsx_agg_update_coded_blocks_sized(st._h, nb, a_rows, a_cols, a_dicts, a_entries, None, None)
        assert rc == 0, f"qsx_agg_update_coded_blocks_sized: {rc}"
    st.clear()
    update()
    torch.cuda.synchronize()
    ms = launches_ms(update, warm=1)
    st.clear()
    update()
    bk, bv, _, bg = st.finalize(dev, capacity=16)
    checked = False
    if not args.no_check:
        ck, cv, cg = single
        g = int(cg.item())
        assert int(bg.item()) == g, "groups of the run of blocks differ from the single stripe's"
        ob = torch.argsort(bk[0][:g].long() * 256 + bk[1][:g].long())
        oc = torch.argsort(ck[0][:g].long() * 256 + ck[1][:g].long())
        for a in range(len(cv)):
            x, y = bv[a][:g][ob], cv[a][:g][oc]
            if a in (0, 7):
                assert bool((x == y).all()), f"aggregate {a} over the run of blocks is not exact"
            else:
                assert bool(torch.allclose(x.double(), y.double(), rtol=1e-9, atol=0.0)), f"aggregate {a} over the run of blocks differs"
        checked = True
    st.close()
    return {"blocks": nb, "rows_per_block": rows_per_block, "ms": ms, "frac": hbm_roofline("", 13 * n, ms)["frac"], "checked": checked,
            "call": "qsx_agg_update_coded_blocks_sized: one launch of factored_coef_kernel (every block's coefficient tables) and one of "
                    "agg_factored_direct_kernel<...,runs>"}


def cpu_baseline_q1_coded(args, threads):
    from oracle import pyoracle as O
    cfg = q1_coded_config()
    block = 4 * 1024 * 1024 // 13
    rng = np.random.default_rng(4)

    def gen(n):
        combo = rng.choice(4, size=n, p=[0.2466, 0.0065, 0.5005, 0.2464])
        return [np.frombuffer(b"ANNR", dtype=np.uint8)[combo], np.frombuffer(b"FFOF", dtype=np.uint8)[combo],
                rng.integers(0, 50, size=n).astype(np.uint8), np.round(rng.uniform(900, 105000, size=n), 2),
                rng.integers(0, 11, size=n).astype(np.uint8), rng.integers(0, 9, size=n).astype(np.uint8)]
    dicts = [None, None, np.arange(1, 51, dtype=np.float64), None, np.arange(11) / 100.0, np.arange(9) / 100.0]

    def run(cols, n):
        secs, st = O.bench_agg_coded(cfg, cols, dicts, n, block, threads)
        st.close()
        return secs
    small = gen(1_000_000)
    n = sized_sample(lambda: run(small, 1_000_000), 1_000_000, args.secondary_cpu_seconds / 5.0, 2_000_000, 40_000_000)
    cols = gen(n)
    secs, trials = trials_2_to_4(lambda: run(cols, n))
    return {"value": n / secs, "unit": "rows/s", "cores": threads, "kind": "port", "trials_s": trials,
            "sample": f"oracle: ThreadPrivateCompactKey aggregation over CompressedColumnStore blocks (per-value dictionary decode, "
                      f"CompressedColumnStoreValueAccessor.hpp:90-150), {n} rows in 4 MB blocks, {threads} worker threads; 5 trials, mean of 2-4"}


def secondary_c3_minimal(ctx, args, threads):
    """BASELINE config 3 as its one-line description reads: SUM / COUNT / AVG GROUP BY 2 keys — two INT keys (3 x 3 groups)
    and one DOUBLE, 16 B/row (SURVEY.md §8(d) "minimal variant")."""
    dev, n = ctx.dev, args.agg_rows
    g = torch.Generator(device=dev)
    g.manual_seed(9)
    k1 = torch.randint(0, 3, (n,), device=dev, generator=g, dtype=torch.int32)
    k2 = torch.randint(0, 3, (n,), device=dev, generator=g, dtype=torch.int32)
    val = torch.rand(n, device=dev, generator=g, dtype=torch.float64)
    st = capi.AggState(minimal_config())
    cols = [k1, k2, val]

    def one():
        st.clear()
        st.update(cols, n)
    ms_clear_update = launches_ms(one)
    ms = launches_ms(lambda: st.update(cols, n), warm=0)
    st.clear()
    st.update(cols, n)
    keys, vals, _, groups = st.finalize(dev, capacity=16)
    checked = False
    if not args.no_check:
        gcount = int(groups.item())
        assert gcount == 9, gcount
        combo = keys[0][:gcount].long() * 3 + keys[1][:gcount].long()
        want_count = torch.bincount(k1.long() * 3 + k2.long(), minlength=9)
        assert bool((vals[1][:gcount] == want_count[combo]).all()), "COUNT(*) per group differs from a bincount of the keys"
        for i in range(gcount):
            mask = (k1 == keys[0][i]) & (k2 == keys[1][i])
            ref = torch.where(mask, val, torch.zeros((), dtype=torch.float64, device=dev)).sum().item()
            assert abs(vals[0][i].item() - ref) <= 1e-6 * abs(ref), "SUM per group differs from a torch reduction"
            assert abs(vals[2][i].item() - ref / want_count[combo[i]].item()) <= 1e-6 * abs(ref), "AVG per group differs"
            del mask
        checked = True
    st.close()
    out = {"workload": f"C3 minimal variant: SUM / COUNT / AVG GROUP BY 2 INT keys over {n} rows, 16 B/row",
           "ms": ms, "ms_with_clear": ms_clear_update, "rows_per_s": n / ms * 1e3,
           "roofline": hbm_roofline("agg_hash_shape_fixed_kernel<ShapeTwoIntKeysSumCountAvg,...> (qsx_agg_update; AOT plan shape)",
                                    16 * n, ms, algorithmic_bytes_per_row=16),
           "checked": checked}
    del k1, k2, val
    if not args.no_cpu_baseline:
        from oracle import pyoracle as O
        cfg = minimal_config()
        rng = np.random.default_rng(9)

        def gen(m):
            return [rng.integers(0, 3, size=m).astype(np.int32), rng.integers(0, 3, size=m).astype(np.int32), rng.random(m)]

        def run(c, m):
            secs, s_ = O.bench_agg(cfg, c, m, 4 * 1024 * 1024 // 16, threads)
            s_.close()
            return secs
        small = gen(1_000_000)
        m = sized_sample(lambda: run(small, 1_000_000), 1_000_000, args.secondary_cpu_seconds / 5.0, 2_000_000, 60_000_000)
        c = gen(m)
        secs, trials = trials_2_to_4(lambda: run(c, m))
        out["cpu_baseline"] = {"value": m / secs, "unit": "rows/s", "cores": threads, "kind": "port", "trials_s": trials,
                               "sample": f"oracle: ThreadPrivateCompactKey aggregation, {m} rows in 4 MB blocks, {threads} worker threads; 5 trials, mean of 2-4"}
    return out


def secondary_agg_many_groups(ctx, args, threads):
    """The reference's default group-by path at scale (PackedPayloadHashTable behind AggregationOperationState's partitioned
    aggregation, storage/AggregationOperationState.cpp:548-614): COUNT(*) + SUM(double) GROUP BY one INT key with 10^6 distinct
    random keys, 100 M rows, 12 B/row.  Two partition passes on digits of the mixing hash, then 4096 pieces through workgroup-private
    LDS tables (csrc/agg_pieces.hpp); `one_pass_ms` = round 5's path (one partition pass, then NS + 1 global atomics per row)."""
    dev, n, groups = ctx.dev, 100_000_000, 1_000_000
    g = torch.Generator(device=dev)
    g.manual_seed(17)
    key = torch.randint(0, groups, (n,), device=dev, generator=g, dtype=torch.int32)
    val = torch.randint(0, 1 << 20, (n,), device=dev, generator=g, dtype=torch.int32).double() / 64.0    # (sums exact in any order)
    cfg = T.make_agg_config(T.AGG_GENERIC, [(T.INT, None), (T.DOUBLE, None)], keys=[0], aggs=[(T.AGG_COUNT_STAR, None), (T.AGG_SUM, T.col(1))],
                            est_groups=groups)
    res = {}
    for name, env in (("two_level", None), ("one_pass", "0")):
        if env is not None:
            os.environ["QSX_AGG_TWO_LEVEL_MIN_GROUPS"] = env
        st = capi.AggState(cfg)

        def one():
            st.clear()
            st.update([key, val], n)
        res[name] = launches_ms(one)
        os.environ.pop("QSX_AGG_TWO_LEVEL_MIN_GROUPS", None)
        if name == "two_level":
            checked = False
            if not args.no_check:
                keys, vals, _, found = st.finalize(dev)
                k = int(found.item())
                want_count = torch.bincount(key.long(), minlength=groups)
                want_sum = torch.zeros(groups, dtype=torch.float64, device=dev).index_add_(0, key.long(), val)
                assert k == int((want_count > 0).sum().item()), "number of groups differs from the distinct keys"
                gk = keys[0][:k].long()
                assert bool((vals[0][:k] == want_count[gk]).all()), "COUNT(*) per group differs from a bincount of the keys"
                assert bool((vals[1][:k] == want_sum[gk]).all()), "SUM per group differs from an index_add of the values"
                checked = True
                del want_count, want_sum, keys, vals
        st.close()
    ms = res["two_level"]
    out = {"workload": f"COUNT(*) + SUM(double) GROUP BY one INT key, {groups} random groups over {n} rows, 12 B/row (state cleared inside the timed call)",
           "ms": ms, "one_pass_ms": res["one_pass"], "rows_per_s": n / ms * 1e3,
           "roofline": hbm_roofline("partition_scatter_kernel<PackedKey,4|3> x 2 + agg_pieces_kernel<1> (qsx_agg_update: two partition passes, "
                                    "4096 pieces through LDS tables)", 12 * n, ms, algorithmic_bytes_per_row=12,
                                    note="three kernels move the rows: 5 x the algorithmic bytes cross HBM (12 read + 12 written per pass, 12 read by the pieces)"),
           "checked": checked}
    del key, val
    if not args.no_cpu_baseline:
        from oracle import pyoracle as O
        rng = np.random.default_rng(17)

        def gen(m):
            return [rng.integers(0, groups, size=m).astype(np.int32), rng.integers(0, 1 << 20, size=m).astype(np.float64) / 64.0]

        def run(c, m):
            secs, s_ = O.bench_agg(cfg, c, m, 4 * 1024 * 1024 // 12, threads)
            s_.close()
            return secs
        small = gen(1_000_000)
        m = sized_sample(lambda: run(small, 1_000_000), 1_000_000, args.secondary_cpu_seconds / 5.0, 2_000_000, 40_000_000)
        c = gen(m)
        secs, trials = trials_2_to_4(lambda: run(c, m))
        out["cpu_baseline"] = {"value": m / secs, "unit": "rows/s", "cores": threads, "kind": "port", "trials_s": trials,
                               "sample": f"oracle: PackedPayloadHashTable aggregation, {groups} groups, {m} rows in 4 MB blocks, {threads} worker threads; 5 trials, mean of 2-4"}
    return out


def secondary_join_small_build(ctx, args, threads):
    """C2 with a build side that fits LDS (a dimension table: 25 K keys — nation / region / a filtered dimension / one partition
    of a radix split): north_star's "LDS-staged hash tables".  Every workgroup copies the table into its LDS and answers its
    probe rows from there (csrc/join_lds.hpp); algorithmic bytes 4 N + 8 M as for C2."""
    dev = ctx.dev
    n_build, n = 25_000, args.probe_rows
    g = torch.Generator(device=dev)
    g.manual_seed(12)
    build = torch.randperm(n_build, device=dev, generator=g, dtype=torch.int32)
    probe = torch.randint(0, n_build, (n,), device=dev, generator=g, dtype=torch.int32)
    out = (torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int64, device=dev))
    res = {}
    for name, key_range in (("dense", (0, n_build - 1)), ("hashed", None)):
        t = capi.JoinTable(T.INT, n_build, key_range=key_range)
        t.build(build)
        ms = launches_ms(lambda: t.probe(probe, capacity=n, out=out))
        k = int(out[2].item())
        checked = False
        if not args.no_check:
            check_pairs(probe, build, out[0], out[1], k, n_build)
            assert k == n
            checked = True
        res[name] = {"ms": ms, "checked": checked}
        t.close()
    ms = res["dense"]["ms"]
    out_line = {"workload": f"C2 with a build side that fits LDS: {n_build} x {n} INTEGER inner equi-join, match rate 1.0",
                "ms": ms, "rows_per_s": n / ms * 1e3, "hashed_table_ms": res["hashed"]["ms"],
                "roofline": hbm_roofline("lds_dense_probe_kernel<int,0> (qsx_join_probe: the table copied into every workgroup's LDS)", 12 * n, ms,
                                         algorithmic_bytes="4*N_probe + 8*N_match"),
                "checked": res["dense"]["checked"] and res["hashed"]["checked"]}
    del out, probe
    if not args.no_cpu_baseline:
        from oracle import pyoracle as O
        rng = np.random.default_rng(12)
        b = rng.permutation(n_build).astype(np.int32)
        small = rng.integers(0, n_build, size=2_000_000).astype(np.int32)
        m = sized_sample(lambda: O.bench_join(b, small, 1_048_576, threads)["probe_seconds"], 2_000_000, args.secondary_cpu_seconds / 5.0, 4_000_000, n)
        p = rng.integers(0, n_build, size=m).astype(np.int32)
        secs, trials = trials_2_to_4(lambda: O.bench_join(b, p, 1_048_576, threads)["probe_seconds"])
        out_line["cpu_baseline"] = {"value": m / secs, "unit": "rows/s", "cores": threads, "kind": "port", "trials_s": trials,
                                    "sample": f"oracle: SimpleScalarSeparateChaining probe, {n_build} x {m} rows in 4 MB blocks, {threads} worker threads; 5 trials, mean of 2-4"}
    return out_line


def secondary_c1_select(ctx, args, threads):
    """BASELINE config 1: SelectOperator over a 10 M-row INTEGER column, predicate col < K.  The reference's own CPU-runnable case is
    the headline there (rows/s through the CPU WorkOrder path: quickstep_amd/host's on_gpu = false work orders under Foreman + Workers,
    tests/cpp/select_cpu_workorder_test — plumbing, no GPU); beside it the same selects on the device: K1 (bitmap) + K2 (compaction)
    through the C ABI on the same column shape, algorithmic bytes 4 N read + 4 sigma N written."""
    import re
    import subprocess
    dev, n = ctx.dev, 10_000_000
    out = {"workload": f"C1 SelectOperator: {n}-row INTEGER column, predicate col < K at selectivities 0.01 / 0.1 / 0.5"}
    exe = os.path.join(ROOT, "tests", "cpp", "bin", "select_cpu_workorder_test")
    workers = min(threads, 16)
    cpu = {}
    if os.path.exists(exe):
        r = subprocess.run([exe, str(n), str(workers)], capture_output=True, text=True, timeout=600)
        for m in re.finditer(r"select col < (\d+): \d+ rows, (\d+) selected, \d+ workers, ([0-9.]+) ms", r.stdout):
            cpu[round(int(m.group(2)) / n, 2)] = {"ms": float(m.group(3)), "rows_per_s": n / float(m.group(3)) * 1e3, "selected": int(m.group(2))}
        out["cpu_workorder_path"] = {"workers": workers, "checked": r.returncode == 0 and "[  PASSED  ]" in r.stdout, "by_selectivity": cpu,
                                     "what": "SelectWorkOrder::executeOnHost over 1 Mi-row blocks under ForemanSingleNode + Workers (BASELINE.md C1: plumbing, no GPU)"}
    else:
        out["cpu_workorder_path"] = {"error": "tests/cpp/bin/select_cpu_workorder_test is not built"}
    g = torch.Generator(device=dev)
    g.manual_seed(21)
    col = torch.randint(0, 2**31 - 1, (n,), device=dev, generator=g, dtype=torch.int32)
    gpu = {}
    checked = True
    for sel in (0.01, 0.1, 0.5):
        k = int(sel * (2**31 - 1))

        def one():
            bm, _ = capi.select_cmp(col, T.LT, k)
            return capi.compact_gather([col], bm, n)
        ms = launches_ms(one)
        (vals,), cnt = one()
        c = int(cnt.item())
        if not args.no_check:
            want = col[col < k]
            checked = checked and c == want.numel() and bool((vals[:c] == want).all())
        gpu[sel] = {"ms": ms, "rows_per_s": n / ms * 1e3, "selected": c,
                    "roofline": hbm_roofline("select_cmp_kernel + compact_gather kernels (qsx_select_cmp, qsx_compact_gather)", 4 * n + 4 * c, ms,
                                             algorithmic_bytes="4 N + 4 sigma N")}
    out.update({"gpu_k1_k2": gpu, "ms": gpu[0.1]["ms"], "roofline": gpu[0.1]["roofline"], "checked": checked and out["cpu_workorder_path"].get("checked", False),
                "note": "10 M rows are 40 MB: three launches of a few microseconds each; the rate is launch-bound at this size, the kernels' own "
                        "rates at 100 M rows are in profiles/r05_bench_ops.jsonl"})
    if cpu:
        out["cpu_baseline"] = {"value": cpu.get(0.1, next(iter(cpu.values())))["rows_per_s"], "unit": "rows/s", "cores": workers, "kind": "port",
                               "sample": f"the whole configuration: {n} rows through the CPU SelectWorkOrder path, selectivity 0.1"}
    return out


def condensed(line):
    """What a partitioned configuration's own line (--config c4 | c5) contributes to `secondary`."""
    return {"workload": line["config"]["workload"], "ms": line["ms_per_step"], "rows_per_s": line["value"],
            "roofline": line["roofline"], "checked": line["checked"], "phases_ms": line["phases_ms"],
            "config": {k: v for k, v in line["config"].items() if k != "workload"}}


def partitioned_operators_leg(args, config, size, twin_ms):
    """BASELINE config 4 / 5 through the operator boundary: the plan's DAG as RelationalOperators under ForemanSingleNode + Workers, the
    exchange steps issued by PartitionExchangeOperator / ExchangeAggregationStatesOperator over a one-rank RankGroup on RCCL
    (tests/cpp/partitioned_operators_bench.cpp, a child process with its own copy of the relations, result-checked every step)."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "cpp", "bin", "partitioned_operators_bench")
    if not os.path.exists(exe):
        return {"error": "tests/cpp/bin/partitioned_operators_bench is not built (make -C quickstep_amd/host)"}
    cmd = [exe, config, str(size), str(args.secondary_steps), "3", str(args.operator_workers), str(args.partitioned_blocks_per_work_order)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    except subprocess.TimeoutExpired:
        return {"error": "timed out"}
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": f"exit code {r.returncode}", "stderr": r.stderr[-1500:]}
    out = json.loads(lines[-1])
    out["ms_over_the_raw_abi_plan"] = out["ms_per_step"] / twin_ms
    return out


def secondary_c4(ctx, args, threads):
    from quickstep_amd import plans
    sub = argparse.Namespace(**vars(args))
    sub.steps, sub.warmup = args.secondary_steps, 2
    out = condensed(run_c4(ctx, sub))
    if not args.no_operators:
        torch.cuda.empty_cache()
        out["operators"] = partitioned_operators_leg(args, "c4", args.c4_orders_per_rank, out["ms"])
    if not args.no_cpu_baseline:
        from oracle import pyoracle as O
        P, block = 8, 4 * 1024 * 1024 // 12      # SF100 over 8 partitions; 4 MB blocks of (INT key, 8-byte payload)

        def host(n_o):
            inp = plans.generate_c4_inputs(ctx.dev, n_o, 0)
            return [inp[k].cpu().numpy() for k in ("o_orderkey", "o_payload", "l_orderkey", "l_payload")]

        def run(cols):
            r = O.bench_partitioned_join(cols[0], cols[1], cols[2], cols[3], P, block, threads)
            assert r["violations"] == 0 and r["output_rows"] == cols[2].size
            return r["repartition_seconds"] + r["build_seconds"] + r["probe_seconds"], r
        small = host(500_000)
        n_o = sized_sample(lambda: run(small)[0], 500_000, args.secondary_cpu_seconds / 5.0, 1_000_000, args.c4_orders_per_rank)
        cols = host(n_o)
        last = {}

        def trial():
            secs, r = run(cols)
            last.update(r)
            return secs
        secs, trials = trials_2_to_4(trial)
        rows = cols[0].size + cols[2].size
        out["cpu_baseline"] = {"value": rows / secs, "unit": "rows/s", "cores": threads, "kind": "port", "trials_s": trials,
                               "phases_s": {k: last[k] for k in ("repartition_seconds", "build_seconds", "probe_seconds")},
                               "sample": f"oracle: both relations through a repartitioning Select into {P} hash partitions "
                                         f"(PartitionSchemeHeader.hpp:200-214), per-partition SimpleScalarSeparateChaining build + probe + "
                                         f"materialised (key, o_payload, l_payload) (HashJoinOperator.cpp:220-231, 450-541), one address "
                                         f"space; {cols[0].size} orders + {cols[2].size} lineitems, {threads} worker threads; 5 trials, mean of 2-4"}
    return out


def secondary_c5(ctx, args, threads):
    from quickstep_amd import plans
    sub = argparse.Namespace(**vars(args))
    sub.steps, sub.warmup = args.secondary_steps, 2
    out = condensed(run_c5(ctx, sub))
    if not args.no_operators:
        torch.cuda.empty_cache()
        out["operators"] = partitioned_operators_leg(args, "c5", args.c5_sf_per_rank, out["ms"])
    if not args.no_cpu_baseline:
        from oracle import pyoracle as O

        def host(sf):
            inp = plans.generate_q3_inputs(ctx.dev, sf, 0, 1)
            return {k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in inp.items()}

        def run(inp):
            r = O.bench_q3(inp, plans.SEG_BUILDING, plans.DATE_CUT, 262144, threads)
            return r["total_seconds"], r
        small = host(0.5)
        sf = sized_sample(lambda: run(small)[0], 500, args.secondary_cpu_seconds / 5.0, 1000, int(args.c5_sf_per_rank * 1000)) / 1000.0
        inp = host(sf)
        last = {}

        def trial():
            secs, r = run(inp)
            last.update(r)
            return secs
        secs, trials = trials_2_to_4(trial)
        rows = inp["c_custkey"].size + inp["o_orderkey"].size + inp["l_orderkey"].size
        out["cpu_baseline"] = {"value": rows / secs, "unit": "rows/s", "cores": threads, "kind": "port", "trials_s": trials,
                               "phases_s": {k: last[k] for k in ("customer_seconds", "orders_seconds", "lineitem_seconds", "finalize_seconds")},
                               "groups": last["groups"], "pairs": last["pairs"],
                               "sample": f"oracle: Q3 in one process at SF {sf} ({rows} input rows): selects, exact LIP bit vectors on "
                                         f"custkey / orderkey, two SimpleScalarSeparateChaining joins, CollisionFreeVector SUM on l_orderkey, "
                                         f"top 10; block-at-a-time, {threads} worker threads; 5 trials, mean of 2-4"}
    return out


def secondary_block(ctx, args):
    """Runs after the headline's timed region and checks, in the same process: the driver's clock covers it."""
    t_start = time.perf_counter()
    threads = usable_cores()
    out = {}
    legs = [("c1_select", secondary_c1_select), ("q1_coded", secondary_q1_coded), ("c3_minimal", secondary_c3_minimal),
            ("join_small_build", secondary_join_small_build), ("agg_many_groups", secondary_agg_many_groups)]
    for name, fn in legs:
        t0 = time.perf_counter()
        try:
            out[name] = fn(ctx, args, threads)
        except Exception as exc:  # noqa: BLE001  (a secondary leg never takes the headline line down; its failure is in the line)
            out[name] = {"error": repr(exc)[:400], "checked": False}
        out[name]["wall_s"] = time.perf_counter() - t0
    ctx.headline = None          # the 20 GB of Q1 columns are not needed any more
    torch.cuda.empty_cache()
    # C4 / C5 are written against a process group at any N: one rank over RCCL here
    group_made = False
    try:
        if not dist.is_initialized():
            import socket
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                port = sock.getsockname()[1]
            os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": "0", "WORLD_SIZE": "1"})
            dist.init_process_group(backend="nccl", device_id=ctx.dev)
            group_made = True
        sub = Ctx()
        sub.dev, sub.rank, sub.world, sub.distributed, sub.group = ctx.dev, 0, 1, True, None
        for name, fn in (("c4", secondary_c4), ("c5", secondary_c5)):
            t0 = time.perf_counter()
            try:
                out[name] = fn(sub, args, threads)
            except Exception as exc:  # noqa: BLE001
                out[name] = {"error": repr(exc)[:400], "checked": False}
            out[name]["wall_s"] = time.perf_counter() - t0
            torch.cuda.empty_cache()
    finally:
        if group_made:
            dist.destroy_process_group()
    out["wall_s"] = time.perf_counter() - t_start
    return out


def operators_leg(args, raw_value):
    """The same workload through the operator boundary: BuildHash / HashJoin / Aggregation / FinalizeAggregation operators
    of quickstep_amd/host under ForemanSingleNode + Workers on reference-sized 4 MB blocks, work orders over runs of
    blocks (tests/cpp/headline_operators_bench.cpp, a child process with its own copy of the relations).  Reported next to
    the raw-ABI value, never instead of it."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "cpp", "bin", "headline_operators_bench")
    if not os.path.exists(exe):
        return {"error": "tests/cpp/bin/headline_operators_bench is not built (make -C quickstep_amd/host)"}
    def child(lineitem_store):
        # (the compressed relations take twice the blocks per work order: five or six aggregation work orders per step instead
        # of ten — every one of them scans its predicate, builds its coefficient tables and settles its cells once)
        blocks = args.blocks_per_work_order * (2 if lineitem_store != 0 else 1)
        # (ten untimed steps at least: every Worker thread keeps a cache of device scratch of its own, which is warm only once
        # the thread has executed each kind of work order — the first handful of steps carry 6-9 ms of allocations now and then)
        cmd = [exe, str(args.build_rows), str(args.probe_rows), str(args.agg_rows), str(args.steps), str(max(args.warmup, 10)),
               str(args.operator_workers), str(blocks), str(lineitem_store)]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        except subprocess.TimeoutExpired:
            return {"error": "timed out"}
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": f"exit code {r.returncode}", "stderr": r.stderr[-1500:]}
        return json.loads(lines[-1])
    out = child(0)
    if "error" in out:
        return out
    # the same step with lineitem as the reference's DDL stores it: CompressedColumnStore block images in the reference's layout,
    # adopted in place; the aggregation reads 13 B/row and its aggregates are factored through every block's own dictionaries
    coded = child(1)
    if "rows_per_s" in coded:
        coded["fraction_of_raw_abi_value"] = coded["rows_per_s"] / raw_value
    out["compressed_lineitem"] = coded
    # and with TPC-H Q1's WHERE l_shipdate <= DATE inside the aggregation, l_shipdate one more dictionary-coded attribute (2-byte
    # codes, a dictionary per block) of images sorted on l_orderkey: the predicate is scanned on the code stripes, rewritten on
    # every block's own dictionary
    with_predicate = child(2)
    if "rows_per_s" in with_predicate:
        with_predicate["fraction_of_raw_abi_value"] = with_predicate["rows_per_s"] / raw_value
    out["compressed_lineitem_q1_predicate"] = with_predicate
    out["fraction_of_raw_abi_value"] = out["rows_per_s"] / raw_value
    out["note"] = ("the operators' join produces its output relation (one INT attribute from each side, written by the probe: "
                   "qsx_join_probe_project_blocks) where the raw-ABI step stops at the (probe_tid, build_tid) pairs: "
                   "fraction_of_raw_abi_with_output_relation compares with the raw step whose probe writes that relation too "
                   "(probe.variants.dense_m1.0_projected_int_attributes); fraction_of_raw_abi_plus_materialisation with the raw step + "
                   "two gathers by the pair list (dense_m1.0_materialised - dense_m1.0: the unfused form)")
    return out


def capi_leg_under_deadline(ctx, args, run_config, torch_line, legs):
    """The second leg of --transport both: the same configuration with every exchange step through the C ABI's collectives.
    Returns the line to print: the capi leg's when it completed and checked on EVERY rank, else the torch leg's with the reason.
    A stall (a rank inside a collective its peers never enter) ends at the deadline: rank 0 prints the torch line from the
    watchdog thread and every rank leaves with status 0 — os._exit, because a process stuck in a collective cannot unwind."""
    import threading
    from quickstep_amd import distributed as qd
    os.environ.setdefault("QSX_COMM_TIMEOUT_MS", str(int(args.capi_leg_deadline * 500)))   # the library's own watchdog: half the deadline
    done = threading.Event()

    def fallback(reason):
        legs["capi"] = {"error": reason}
        torch_line["transport_legs"] = legs
        torch_line["world_size_seen"] = ctx.world
        torch_line["self_launched"] = os.environ.get("QSX_BENCH_SELF_LAUNCHED") == "1"
        return torch_line

    def watchdog():
        if done.wait(args.capi_leg_deadline):
            return
        if ctx.rank == 0:
            import ctypes
            ctypes.CDLL(None).fflush(None)
            print(json.dumps(fallback(f"the C ABI leg did not finish within {args.capi_leg_deadline} s")), flush=True)
        os._exit(0)
    threading.Thread(target=watchdog, daemon=True).start()
    ok, error, line = 1, None, None
    try:
        broken = os.environ.get("QSX_BENCH_BREAK_CAPI_LEG")     # tests: "all" = every rank fails alike; "last" = one rank fails, its peers stall
        if broken == "all" or (broken == "last" and ctx.rank == ctx.world - 1):
            raise RuntimeError("QSX_BENCH_BREAK_CAPI_LEG")
        ctx.group = qd.CapiGroup.from_torch_group(capi, ctx.dev)
        line = run_config(ctx, args)
    except BaseException as exc:  # noqa: BLE001  (this leg never takes the run down)
        ok, error = 0, repr(exc)[:400]
    try:
        if ctx.group is not None:
            ctx.group.comm.close()
    except Exception:  # noqa: BLE001
        pass
    ctx.group = None
    flag = torch.tensor([ok], dtype=torch.int32, device=ctx.dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    everywhere = ok == 1 and int(flag.item()) == 1
    done.set()
    if not everywhere:
        return fallback(error or "the C ABI leg failed on another rank")
    line["transport"] = "capi"
    legs["capi"] = {k: line.get(k) for k in ("value", "ms_per_step") if k in line}
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--c4-unfused", action="store_true", help="config c4: pair list + K5 gathers instead of the projecting probe")
    ap.add_argument("--c4-no-overlap", action="store_true", help="config c4: the probe side's exchange after the build instead of next to it")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=["headline", "c4", "c5"], default="headline")
    ap.add_argument("--build-rows", type=int, default=1_000_000)
    ap.add_argument("--probe-rows", type=int, default=100_000_000)
    ap.add_argument("--agg-rows", type=int, default=600_000_000)
    ap.add_argument("--match", type=float, default=1.0)
    ap.add_argument("--c4-orders-per-rank", type=int, default=18_750_000, help="150 M orders of SF100 over 8 ranks")
    ap.add_argument("--c5-sf-per-rank", type=float, default=37.5, help="SF300 over 8 ranks")
    ap.add_argument("--no-lip", action="store_true", help="c5: no LIP filters")
    ap.add_argument("--join-table", choices=["dense", "hashed"], default="dense",
                    help="dense: the build key (custkey) has exact min/max statistics -> directly addressed table "
                         "(qsx_join_table_create_dense); hashed: open-addressing table (qsx_join_table_create)")
    ap.add_argument("--join-plan", choices=["auto", "shuffle", "broadcast"], default="auto",
                    help="N > 1: shuffle = both sides repartitioned on the join key (K9 + RCCL all-to-all); broadcast = "
                         "all-gather of the build side, probe rows stay where they are (the reference's broadcast join, "
                         "BuildHashOperator.hpp:99,146-152); auto = shuffle (the exchange BASELINE.json names)")
    ap.add_argument("--other-plan-leg", action="store_true",
                    help="N > 1: after the timed region also run the join plan that was NOT picked, untimed, and report its cost")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU time budget per operator for the cpu_baseline trials")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the result checks after the timed region")
    ap.add_argument("--no-probe-variants", action="store_true", help="N = 1: skip the other C2 legs (PMC passes: only the headline's own launches)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="N = 1: skip the `secondary` block (Q1 over code stripes, C3 minimal, C4, C5 with their cpu baselines)")
    ap.add_argument("--secondary-steps", type=int, default=5, help="timed steps of the C4 / C5 legs of `secondary`")
    ap.add_argument("--secondary-cpu-seconds", type=float, default=4.0, help="CPU time budget per `secondary` cpu_baseline (5 trials in all)")
    ap.add_argument("--no-operators", action="store_true", help="N = 1: skip the leg that runs the workload through the C++ operator layer")
    ap.add_argument("--operator-workers", type=int, default=4,
                    help="Worker threads of the operators legs (8 until late in round 5: every Worker spins in its stream wait, and on a box "
                         "that grants 16 cores by cgroup quota eight of them next to the runtime's own threads got the process throttled — "
                         "6-7 ms standstills every few steps; four do the same work in the same or less time)")
    ap.add_argument("--blocks-per-work-order", type=int, default=256)
    ap.add_argument("--partitioned-blocks-per-work-order", type=int, default=1024,
                    help="blocks per work order of the C4 / C5 operators legs (their work orders stream: every one ends in a count the host "
                         "reads, so fewer and longer ones)")
    ap.add_argument("--transport", choices=["auto", "torch", "capi", "both"], default="auto",
                    help="who issues the exchange steps: torch = torch.distributed on the nccl backend (RCCL); capi = the C ABI's own "
                         "multi-GPU entry points (qsx_alltoallv, qsx_allgather, qsx_agg_reduce_scatter, ...: RCCL bound inside "
                         "libqsx.so, the calls the C++ operator layer makes — quickstep_amd/host/rank_exchange.cpp); both = the torch leg, "
                         "then the capi leg under a deadline, the line's value from the capi leg when it completed on every rank; "
                         "auto = both at N > 1, torch with one rank")
    ap.add_argument("--capi-leg-deadline", type=float, default=240.0,
                    help="--transport both: seconds the C ABI leg may take before the line falls back to the torch leg")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch check without a GPU: the ranks rendezvous over gloo, agree on the world size and rank 0 prints a "
                         "line with no measurement in it (tests/test_bench_launch.py)")
    args = ap.parse_args()

    ctx = Ctx()
    ctx.world = world = int(os.environ.get("WORLD_SIZE", "1"))
    ctx.rank = rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} was launched with WORLD_SIZE={world}: the rank count must equal --gpus")
    if args.dry_run:
        if world > 1:
            dist.init_process_group(backend="gloo")
            seen = torch.ones(1, dtype=torch.int64)
            dist.all_reduce(seen)
            world_seen = int(seen.item())
            dist.destroy_process_group()
        else:
            world_seen = 1
        if rank == 0:
            print(json.dumps({"metric": METRIC, "value": None, "dry_run": True, "n_gpus": args.gpus, "world_size_seen": world_seen,
                              "self_launched": os.environ.get("QSX_BENCH_SELF_LAUNCHED") == "1", "config": {"workload": args.config}}), flush=True)
        return
    # QSX_BENCH_SHARED_GPU=1: a rehearsal of the N > 1 path on a box with ONE GPU — every rank works on cuda:0, the process
    # group is gloo (barriers, the max over ranks, the checks) and the exchange steps go through the C ABI over the loopback
    # stand-in for RCCL (QSX_ALLOW_TEST_TRANSPORT=1 QSX_RCCL_LIBRARY=tests/cpp/bin/libloopback_rccl.so, --transport capi).  It executes every line of the
    # multi-rank code with the product's kernels; its throughput says nothing about scaling, and the line says so.
    shared_gpu = os.environ.get("QSX_BENCH_SHARED_GPU") == "1"
    if shared_gpu:
        if args.transport == "torch" or not os.environ.get("QSX_RCCL_LIBRARY") or os.environ.get("QSX_ALLOW_TEST_TRANSPORT") != "1":
            raise SystemExit("QSX_BENCH_SHARED_GPU=1 needs --transport capi (or both: the torch leg then stages through the host over gloo), "
                             "QSX_RCCL_LIBRARY (the loopback library) and QSX_ALLOW_TEST_TRANSPORT=1")
        local_rank = 0
    torch.cuda.set_device(local_rank)
    ctx.dev = dev = torch.device("cuda", local_rank)
    if capi.device_count() < 1:
        raise SystemExit("libqsx.so sees no gfx950 device; there is no CPU path to benchmark")
    # QSX_BENCH_FORCE_DISTRIBUTED=1 runs the multi-GPU code path (RCCL shuffle + merge) even with one rank:
    # used to validate that path on the 1-GPU box (torch.distributed.run --nproc-per-node 1).
    # (the partitioned configurations are written against a process group at any N: a plain `python bench.py --config c4`
    # makes its own group of one rank)
    ctx.distributed = world > 1 or os.environ.get("QSX_BENCH_FORCE_DISTRIBUTED") == "1" or args.config in ("c4", "c5")
    if ctx.distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if "MASTER_ADDR" not in os.environ:
            import socket
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                port = sock.getsockname()[1]
            os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": "0", "WORLD_SIZE": "1"})
        if shared_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)
    ctx.group = None
    want_secondary = rank == 0 and world == 1 and args.config == "headline" and not args.no_secondary and not ctx.distributed
    ctx.keep_headline = want_secondary
    run_config = {"headline": run_headline, "c4": run_c4, "c5": run_c5}[args.config]
    # Who issues the exchange steps.  auto: one rank -> torch.distributed (nothing is exchanged); N > 1 -> BOTH legs, the C ABI's own
    # collectives (csrc/comm.hip: what a compiled host and the C++ operator layer call) as the line's value and the
    # torch.distributed leg beside it (`transport_legs`).  The torch leg runs first and is kept; the C ABI leg runs under a
    # deadline: RCCL with more than one rank under comm.hip cannot be rehearsed on a one-GPU box, so if that leg raises on any
    # rank or stalls, the line falls back to the torch leg and says why — never a lost run.
    transport = args.transport
    if transport == "auto":
        transport = "capi" if shared_gpu else ("both" if world > 1 else "torch")
    legs = {}
    if transport == "both":
        line = run_config(ctx, args)
        line["transport"] = "torch"
        legs["torch"] = {k: line.get(k) for k in ("value", "ms_per_step", "phases_ms", "alltoall", "collectives") if k in line}
        torch.cuda.empty_cache()
        line = capi_leg_under_deadline(ctx, args, run_config, line, legs)
    else:
        if ctx.distributed and transport == "capi":
            from quickstep_amd import distributed as qd
            ctx.group = qd.CapiGroup.from_torch_group(capi, dev)
        line = run_config(ctx, args)
        line["transport"] = transport if ctx.distributed else None
    if legs:
        line["transport_legs"] = legs
    line["world_size_seen"] = dist.get_world_size() if ctx.distributed else 1      # what the RCCL process group reports
    line["self_launched"] = os.environ.get("QSX_BENCH_SELF_LAUNCHED") == "1"
    if shared_gpu:
        line["rehearsal"] = "ranks share cuda:0 over the loopback transport: the N > 1 code path, not a scaling number"
    if ctx.group is not None:
        ctx.group.comm.close()
    if rank == 0 and world == 1 and args.config == "headline" and not args.no_operators:
        line["operators"] = operators_leg(args, line["value"])
        variants = line.get("probe", {}).get("variants", {})
        if "rows_per_s" in line["operators"] and "dense_m1.0_materialised" in variants and "dense_m1.0" in variants:
            extra_ms = variants["dense_m1.0_materialised"]["ms"] - variants["dense_m1.0"]["ms"]
            raw_ms = line["ms_per_step"] + extra_ms
            line["operators"]["fraction_of_raw_abi_plus_materialisation"] = raw_ms / line["operators"]["ms_per_step"]
        if "rows_per_s" in line["operators"] and "dense_m1.0_projected_int_attributes" in variants:
            # like for like: the raw step with its probe writing the operators' output relation instead of the pair list
            raw_ms = line["ms_per_step"] - line["phases_ms"]["probe"] + variants["dense_m1.0_projected_int_attributes"]["ms"]
            line["operators"]["fraction_of_raw_abi_with_output_relation"] = raw_ms / line["operators"]["ms_per_step"]
    if want_secondary:     # (behind the operators leg: that child process is timed next to an otherwise idle parent, as in round 4)
        line["secondary"] = secondary_block(ctx, args)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.config == "headline":
        line["cpu_baseline"] = cpu_baseline(args)
    # RCCL writes its version banner to the C library's stdout, which a pipe only sees when that buffer is flushed — at
    # exit, behind anything Python printed.  The JSON line is the last thing this process writes: flush C stdio first.
    if ctx.distributed:
        dist.destroy_process_group()
    import ctypes
    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
