"""CPU, world_size 2, gloo: the rank logic of quickstep_amd/distributed.py (join-key shuffle with
split sizes, tid bases, partial-aggregate merge) against a single-process oracle join."""
import os
import socket
import subprocess
import sys

import numpy as np

from quickstep_amd import types as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_partitioned_join_and_merge_two_ranks(tmp_path, oracle):
    world = 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    for prefix in ("rank", "dense_rank", "bcast_rank"):
        _check_partitioned_join(oracle, tmp_path, world, prefix, on_owner_rank=prefix != "bcast_rank")
    _check_uneven_broadcast(oracle, tmp_path, world)
    _check_config4(tmp_path, world)
    _check_config5(tmp_path, world)


def _check_config4(tmp_path, world):
    """(key, o_payload, l_payload) rows of the partitioned join: one per lineitem row, the payloads of the right rows."""
    ranks = [np.load(tmp_path / f"c4_rank{i}.npz") for i in range(world)]
    l_key = np.concatenate([d["l_key"] for d in ranks]).astype(np.int64)
    l_pay = np.concatenate([d["l_pay"] for d in ranks])
    out_key = np.concatenate([d["out_key"] for d in ranks]).astype(np.int64)
    out_o = np.concatenate([d["out_o"] for d in ranks])
    out_l = np.concatenate([d["out_l"] for d in ranks])
    assert out_key.size == l_key.size and np.array_equal(out_o, out_key * 3 + 1)
    a, b = np.stack([out_key, out_l], 1), np.stack([l_key, l_pay], 1)
    assert np.array_equal(a[np.lexsort((a[:, 1], a[:, 0]))], b[np.lexsort((b[:, 1], b[:, 0]))])


def _check_config5(tmp_path, world):
    from helpers import q3_reference_numpy
    ranks = [np.load(tmp_path / f"q3_rank{i}.npz") for i in range(world)]
    inputs = [{k[3:]: d[k] for k in d.files if k.startswith("in_")} for d in ranks]
    keys, sums, pairs = q3_reference_numpy(inputs)
    assert keys.size > 50
    top = np.argsort(-sums, kind="stable")[:10]
    for tag in ("f", "g"):
        assert sum(int(d[f"{tag}_pairs"]) for d in ranks) == pairs
        got_keys = np.concatenate([d[f"{tag}_keys"] for d in ranks])
        got_rev = np.concatenate([d[f"{tag}_rev"] for d in ranks])
        order = np.argsort(got_keys)
        assert np.array_equal(got_keys[order], keys) and np.allclose(got_rev[order], sums, rtol=1e-9, atol=0.0)
        for d in ranks:
            assert np.allclose(d[f"{tag}_top_rev"], sums[top], rtol=1e-9, atol=0.0)
            assert np.array_equal(np.sort(d[f"{tag}_top_keys"]), np.sort(keys[top]))


def _check_uneven_broadcast(oracle, tmp_path, world):
    """Ranks hold different numbers of build rows: global build tid = rank * 5000 + local row (the tid base each rank passed)."""
    ranks = [np.load(tmp_path / f"bcast_uneven_rank{i}.npz") for i in range(world)]
    stride = 5000
    probe = np.concatenate([d["probe_keys"] for d in ranks])
    got = np.concatenate([np.stack([d["pairs_probe"], d["pairs_build"]], 1) for d in ranks])
    want = []
    for r, d in enumerate(ranks):                      # the single-node join, one build slice at a time
        t = oracle.JoinTable(T.INT, d["build_keys"].size)
        t.build(d["build_keys"])
        p, b = t.probe(probe)
        want.append(np.stack([p, b + r * stride], 1))
    want = np.concatenate(want)
    key = lambda a: a[np.lexsort((a[:, 1], a[:, 0]))]  # noqa: E731
    assert np.array_equal(key(got), key(want))


def _check_partitioned_join(oracle, tmp_path, world, prefix, on_owner_rank=True):
    ranks = [np.load(tmp_path / f"{prefix}{i}.npz") for i in range(world)]
    build = np.concatenate([d["build_keys"] for d in ranks])      # global tid = rank * n + local row
    probe = np.concatenate([d["probe_keys"] for d in ranks])
    got = np.concatenate([np.stack([d["pairs_probe"], d["pairs_build"]], 1) for d in ranks])
    # every pair satisfies the join condition and sits on the rank that owns its key's partition
    assert (build[got[:, 1]] == probe[got[:, 0]]).all()
    for i, d in enumerate(ranks):
        keys = probe[d["pairs_probe"]]
        if on_owner_rank:
            assert ((keys.astype(np.uint32) & (world - 1)) == i).all()
        else:   # broadcast join: a pair is produced by the rank that holds the probe row
            n_probe = d["probe_keys"].size
            assert ((d["pairs_probe"] // n_probe) == i).all()
    # and the union over ranks is exactly the single-node join
    t = oracle.JoinTable(T.INT, build.size)
    t.build(build)
    p, b = t.probe(probe)
    want = np.stack([p, b], 1)
    key = lambda a: a[np.lexsort((a[:, 1], a[:, 0]))]  # noqa: E731
    assert np.array_equal(key(got.astype(np.int64)), key(want.astype(np.int64)))


def test_truncated_pair_lists_are_an_error():
    """qsx_join_probe counts past its capacity: materialize() must not hand back the shorter list."""
    import pytest
    import torch
    from quickstep_amd import distributed as qd
    assert qd._checked_count(torch.tensor([4]), torch.empty(4)) == 4
    with pytest.raises(RuntimeError):
        qd._checked_count(torch.tensor([5]), torch.empty(4))
