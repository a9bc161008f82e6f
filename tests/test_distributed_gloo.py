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
