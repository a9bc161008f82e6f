"""BASELINE config 4 on one GPU: the partitioned join's device steps — K9 scatter on the join key for both
sides, one table per partition (strided directly addressed table or hashed), local probe, K5 payload
gathers — run for all P partitions in turn (the exchange itself is covered by the 2-rank gloo test).
The union over the partitions must be the unpartitioned join."""
import numpy as np
import pytest
import torch

from quickstep_amd import types as T
from quickstep_amd.distributed import rank_progression
from helpers import sorted_pairs, to_dev

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P", [2, 8])
@pytest.mark.parametrize("dense", [False, True])
def test_partitioned_join_equals_single_table_join(capi, oracle, dev, P, dense):
    rng = np.random.default_rng(P + 10 * dense)
    n_orders, n_lines = 60_000, 400_000
    o_orderkey = (rng.permutation(n_orders) + 1).astype(np.int32)          # dense, unique
    o_payload = rng.integers(0, 2**40, size=n_orders).astype(np.int64)
    l_orderkey = rng.integers(1, n_orders + 1 + 500, size=n_lines).astype(np.int32)   # some lines have no order
    l_payload = rng.normal(size=n_lines)
    dk_o, dp_o, dk_l, dp_l = (to_dev(a, dev) for a in (o_orderkey, o_payload, l_orderkey, l_payload))
    tid_o = torch.arange(n_orders, dtype=torch.int32, device=dev)
    tid_l = torch.arange(n_lines, dtype=torch.int32, device=dev)
    (bk, bt), boff = capi.partition_scatter(dk_o, P, [dk_o, tid_o])
    (pk, pt), poff = capi.partition_scatter(dk_l, P, [dk_l, tid_l])
    boff, poff = boff.cpu().tolist(), poff.cpu().tolist()
    got_l, got_o, out_lp, out_op = [], [], [], []
    for r in range(P):
        keys_b, keys_p = bk[boff[r]:boff[r + 1]], pk[poff[r]:poff[r + 1]]
        assert bool(((keys_b & (P - 1)) == r).all()) and bool(((keys_p & (P - 1)) == r).all())
        prog = rank_progression((1, n_orders), P, r) if dense else None
        table = capi.JoinTable(T.INT, keys_b.numel(), key_range=prog, key_stride=P if prog else 1)
        table.build(keys_b)
        assert table.size() == keys_b.numel()
        total = int(table.probe_count(keys_p).item())
        p, b, c = table.probe(keys_p, capacity=total)
        assert int(c.item()) == total
        gl = capi.gather(pt[poff[r]:poff[r + 1]], p[:total])                # global lineitem / orders row ids
        go = capi.gather(bt[boff[r]:boff[r + 1]], b[:total])
        got_l.append(gl.cpu().numpy())
        got_o.append(go.cpu().numpy())
        out_lp.append(capi.gather(dp_l, gl).cpu().numpy())                  # materialised payload columns
        out_op.append(capi.gather(dp_o, go).cpu().numpy())
    got_l, got_o = np.concatenate(got_l), np.concatenate(got_o)
    t = oracle.JoinTable(T.INT, n_orders)
    t.build(o_orderkey)
    rp, rb = t.probe(l_orderkey)
    assert np.array_equal(sorted_pairs(got_l, got_o), sorted_pairs(rp, rb))
    assert np.array_equal(np.concatenate(out_lp), l_payload[got_l]) and np.array_equal(np.concatenate(out_op), o_payload[got_o])
    assert np.array_equal(o_orderkey[got_o], l_orderkey[got_l])


def test_distributed_surface_over_rccl():
    """Every collective of quickstep_amd/distributed.py through the real backend (RCCL) with the product's ops, one rank
    (tests/dist_worker_rccl.py): shuffle joins (dense and hashed tables), broadcast join, hash-state merge, dense-state
    reduce-scatter and all-reduce with SUM / MIN / bit-OR."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "dist_worker_rccl.py")]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "RCCL_SURFACE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
