"""bench.py --gpus N started as a plain `python bench.py` (no rank environment) launches its own N ranks as child
processes (VERDICT r03 missing 2).  --dry-run keeps the GPU out of it: the ranks rendezvous over gloo and rank 0 prints
the line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run"] + extra, env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    return json.loads(lines[-1])        # the JSON line is the LAST line of stdout


def test_plain_python_bench_launches_its_ranks():
    for n, cfg in ((2, "headline"), (3, "c4")):
        line = _run(["--gpus", str(n), "--config", cfg, "--steps", "1", "--warmup", "0"])
        assert line["n_gpus"] == n and line["world_size_seen"] == n and line["self_launched"] is True
        assert line["config"]["workload"] == cfg


def test_one_gpu_needs_no_launcher():
    line = _run(["--gpus", "1"])
    assert line["n_gpus"] == 1 and line["world_size_seen"] == 1 and line["self_launched"] is False


def test_under_torch_distributed_run_nothing_is_relaunched():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29611", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["world_size_seen"] == 2 and line["self_launched"] is False


def test_loopback_transport_protocol_selftest():
    """tests/cpp/loopback: the stand-in for RCCL that lets several ranks share one GPU (QSX_RCCL_LIBRARY) — its rendezvous,
    grouped send / recv matching, collectives and reductions at world 1, 2, 3, with host memory in place of device memory."""
    exe = os.path.join(ROOT, "tests", "cpp", "bin", "loopback_selftest")
    assert os.path.exists(exe), "make -C quickstep_amd/host builds it"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "loopback selftest ok" in r.stdout, r.stdout + r.stderr


import pytest  # noqa: E402


def _rehearsal(world, config, transport, sizes):
    lib = os.path.join(ROOT, "tests", "cpp", "bin", "libloopback_rccl.so")
    assert os.path.exists(lib), "make -C quickstep_amd/host builds it"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"QSX_BENCH_SHARED_GPU": "1", "QSX_RCCL_LIBRARY": lib, "QSX_ALLOW_TEST_TRANSPORT": "1"})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--transport", transport, "--config", config,
                        "--steps", "1", "--warmup", "0", "--no-cpu-baseline"] + sizes, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"metric')][-1])
    assert line["n_gpus"] == world and line["world_size_seen"] == world and line["self_launched"] is True
    assert "rehearsal" in line and line["value"] > 0
    assert line["roofline"]["frac"] > 0 and line["scaling"] == "weak"
    return line


@pytest.mark.gpu
@pytest.mark.parametrize("config", ["headline", "c4", "c5"])
def test_two_ranks_rehearsal_on_one_gpu(config):
    """`python bench.py --gpus 2` end to end on the one GPU of the test box: the self-launch, two rank processes, the
    product's kernels, every exchange step of the configuration through the C ABI over the loopback stand-in for RCCL
    (QSX_BENCH_SHARED_GPU=1: both ranks on cuda:0, gloo for barriers and checks).  bench.py checks the results of every
    configuration itself (pair lists, group counts, Q3's top 10 against a torch evaluation) and exits non-zero otherwise."""
    line = _rehearsal(2, config, "capi", [])
    assert line["transport"] == "capi"


# per-rank sizes of the 8-rank rehearsals: eight rank processes share the one GPU and its memory
SMALL = {"headline": ["--build-rows", "200000", "--probe-rows", "8000000", "--agg-rows", "40000000"],
         "c4": ["--c4-orders-per-rank", "1500000"], "c5": ["--c5-sf-per-rank", "3"]}


@pytest.mark.gpu
@pytest.mark.parametrize("config", ["headline", "c4", "c5"])
def test_eight_ranks_rehearsal_on_one_gpu(config):
    """The command the driver's SCALE run issues at N = 8 — `bench.py --gpus 8` with its default transport selection — as eight rank
    processes on the box's one GPU: P = 8 is the partition count of the target node (pid = h & 7,
    catalog/PartitionSchemeHeader.hpp:207-214), an 8-way counts exchange, eight ragged pieces per all-to-all, eight reduce-scatter
    ranges.  The default at N > 1 is --transport both: the torch.distributed leg, then the C ABI's collectives under a deadline; the
    line's value comes from the C ABI leg and carries both."""
    line = _rehearsal(8, config, "both" if config == "headline" else "capi", SMALL[config])   # (auto = both, except under QSX_BENCH_SHARED_GPU)
    assert line["transport"] == "capi"
    if config == "headline":
        assert set(line["transport_legs"]) == {"torch", "capi"} and line["transport_legs"]["torch"]["value"] > 0
        assert "error" not in line["transport_legs"]["capi"]


@pytest.mark.gpu
def test_four_ranks_rehearsal_on_one_gpu():
    # (the torch leg of a rehearsal stages every exchange through the host over gloo: a small scale factor)
    line = _rehearsal(4, "c5", "both", ["--c5-sf-per-rank", "0.4"])
    assert line["transport"] == "capi" and "error" not in line["transport_legs"]["capi"]


@pytest.mark.gpu
@pytest.mark.parametrize("how", ["all", "last"])
def test_a_failing_c_abi_leg_falls_back_to_the_torch_leg(how):
    """--transport both with a C ABI leg that cannot run must still print a line — the torch leg's — that says why: `all` = every
    rank fails alike (a transport that does not load); `last` = one rank fails on its own and its peers wait for it inside the
    leg's first collective: the deadline ends the leg on every rank, rank 0 prints the torch leg's line, all leave with status 0."""
    lib = os.path.join(ROOT, "tests", "cpp", "bin", "libloopback_rccl.so")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"QSX_BENCH_SHARED_GPU": "1", "QSX_RCCL_LIBRARY": lib, "QSX_ALLOW_TEST_TRANSPORT": "1", "QSX_BENCH_BREAK_CAPI_LEG": how})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--transport", "both", "--config", "c4", "--steps", "1",
                        "--warmup", "0", "--no-cpu-baseline", "--capi-leg-deadline", "20"] + SMALL["c4"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"metric')][-1])
    assert line["transport"] == "torch" and "error" in line["transport_legs"]["capi"] and line["value"] > 0
    assert line["n_gpus"] == 2 and line["world_size_seen"] == 2
