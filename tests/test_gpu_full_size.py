"""Parity at BASELINE.json's FULL sizes, inside `-m gpu` (not only in bench.py): size-independent properties, exact for
integer work, 1e-6 relative for DOUBLE sums (north_star).

  C2  1 M x 100 M INTEGER inner join, both table flavours (and the hashed table with its directly addressed shadow switched
      off), match rates 1.0 and 0.2: every pair satisfies the join condition, the probe tids are EXACTLY the probe rows
      whose key has a build row, each once (a permutation of them — sorted equality, not a checksum);
  C3  600 M rows, Q1 shape: COUNT(*) per group equals the number of rows with the group's key, SUM(l_quantity) is exact
      (integer-valued doubles), SUM(l_extendedprice) and SUM(l_extendedprice * (1 - l_discount)) are within 1e-6 of torch's
      f64 reductions, AVG = SUM / COUNT.
"""
import pytest
import torch

from quickstep_amd import types as T

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("flavour", ["dense", "hashed", "hashed_no_shadow"])
def test_c2_full_size_join_pairs_are_exactly_the_matching_probe_rows(capi, dev, flavour, monkeypatch):
    if flavour == "hashed_no_shadow":
        monkeypatch.setenv("QSX_JOIN_ADAPTIVE", "0")
    g = torch.Generator(device=dev)
    g.manual_seed(2)
    n_build, n_probe = 1_000_000, 100_000_000
    build = torch.randperm(n_build, device=dev, generator=g, dtype=torch.int32)
    table = capi.JoinTable(T.INT, n_build, key_range=(0, n_build - 1) if flavour == "dense" else None)
    table.build(build)
    assert table.size() == n_build
    out = (torch.empty(n_probe, dtype=torch.int32, device=dev), torch.empty(n_probe, dtype=torch.int32, device=dev),
           torch.zeros(1, dtype=torch.int64, device=dev))
    for key_space in (n_build, 5 * n_build):                     # match rate 1.0 (foreign key) and 0.2
        g.manual_seed(3)
        probe = torch.randint(0, key_space, (n_probe,), device=dev, generator=g, dtype=torch.int32)
        p, b, cnt = table.probe(probe, capacity=n_probe, out=out)
        k = int(cnt.item())
        want = torch.nonzero(probe < n_build).flatten()
        assert k == want.numel() == int(table.probe_count(probe).item())
        assert bool((build[b[:k].long()] == probe[p[:k].long()]).all())          # the join condition, every pair
        got = torch.sort(p[:k].long()).values
        assert bool((got == want).all())                                         # exactly the matching rows, once each
        del probe, want, got
    table.close()


def test_c3_full_size_q1_aggregates_against_independent_reductions(capi, dev):
    import bench                                                                 # the generator and configuration of the headline
    n = 600_000_000
    cols = bench.gen_q1_columns_gpu(n, dev, 4)
    state = capi.AggState(bench.q1_config())
    state.update(cols, n)
    fin = state.finalize(dev, capacity=16)
    assert bench.check_q1(cols, fin, n) == 4                                     # COUNT / SUM(qty) exact, sums to 1e-6
    keys, vals, nulls, groups = fin
    g = int(groups.item())
    # AVG = SUM / (double) COUNT (expressions/aggregation/AggregationHandleAvg.cpp:144-155)
    assert torch.allclose(vals[4][:g], vals[0][:g] / vals[7][:g].double(), rtol=1e-12, atol=0.0)
    assert torch.allclose(vals[5][:g], vals[1][:g] / vals[7][:g].double(), rtol=1e-12, atol=0.0)
    # the same rows as a run of 4 MB blocks in one launch: identical counts, sums to 1e-6
    rows_per_block = (4 << 20) // 34
    blocks = [[c[a:a + rows_per_block] for c in cols] for a in range(0, 60_000_000, rows_per_block)]
    state.clear()
    state.update_blocks(blocks)
    fin2 = state.finalize(dev, capacity=16)
    rows_total = sum(b[0].numel() for b in blocks)
    assert bench.check_q1([c[:rows_total] for c in cols], fin2, rows_total) == 4


@pytest.mark.parametrize("transport", ["torch", "capi"])
@pytest.mark.parametrize("config", ["c4", "c5"])
def test_c4_c5_at_their_per_rank_full_sizes_over_rccl_one_rank(config, transport):
    """BASELINE configs 4 and 5 at the sizes ONE rank of the 8-GPU job holds (C4: 18.75 M orders + ~75 M lineitems with
    payload columns, both sides through K9 and the exchange; C5: TPC-H Q3 at SF 37.5 with LIP filters, broadcast build sides,
    dense group-by, reduce-scatter) through bench.py's own step and CHECKS — every joined row satisfies the join condition,
    one output row per lineitem row (C4); joined pairs, group count and the top-10 revenues against an independent torch
    evaluation of the query (C5, sums to 1e-6) — over RCCL with one rank, exchanges issued by torch.distributed and by the
    C ABI's own entry points (qsx_alltoallv, qsx_allgather, qsx_agg_reduce_scatter)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", config, "--steps", "1", "--warmup", "0", "--transport", transport],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["world_size_seen"] == 1 and line["transport"] == transport
    if config == "c4":
        assert line["config"]["output_rows"] == line["config"]["lineitems"] > 70_000_000
    else:
        assert line["config"]["joined_pairs"] > 0 and line["config"]["groups"] > 0
