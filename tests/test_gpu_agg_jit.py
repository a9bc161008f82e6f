"""Run-time plan shapes (csrc/agg_jit.hip): the aggregation kernel specialised by hipRTC for the state's
configuration must give the interpreter's results — same body, configuration folded in — on every
strategy, with filters, predicates, MIN/MAX, and when groups overflow LDS.  QSX_AGG_JIT_MIN_ROWS=0 makes
the states compile on their first update (the default waits for 2 Mi rows and compiles in the background)."""
import os

import numpy as np
import pytest

from quickstep_amd import types as T
from helpers import bitmap_dev, to_dev
from test_gpu_agg import FP_RTOL, assert_same_groups, finalize_np, q1_columns, q1_config, run_hip

pytestmark = pytest.mark.gpu


@pytest.fixture
def jit_now(monkeypatch):
    monkeypatch.setenv("QSX_AGG_JIT_MIN_ROWS", "0")
    monkeypatch.delenv("QSX_AGG_NO_SPECIALIZE", raising=False)
    monkeypatch.delenv("QSX_AGG_JIT", raising=False)


def configs(rng, n):
    key = rng.integers(0, 50, size=n).astype(np.int32)
    key2 = rng.integers(0, 300, size=n).astype(np.int32)
    iv = rng.integers(-1000, 1000, size=n).astype(np.int32)
    lv = rng.integers(-2**40, 2**40, size=n).astype(np.int64)
    fv = rng.normal(size=n).astype(np.float32)
    dv = rng.uniform(900, 105000, size=n)
    layout = [(T.INT, None), (T.INT, None), (T.INT, None), (T.LONG, None), (T.FLOAT, None), (T.DOUBLE, None)]
    cols = [key, key2, iv, lv, fv, dv]
    instrs = [(T.EX_MUL, 0, T.col(5), T.col(4)), (T.EX_ADD, 1, T.temp(0), T.col(2)), (T.EX_DIV, 2, T.temp(1), T.const(0)),
              (T.EX_SUB, 0, T.temp(2), T.col(3))]                       # temp 0 is redefined on purpose
    aggs = [(T.AGG_SUM, T.temp(0)), (T.AGG_MIN, T.col(4)), (T.AGG_MAX, T.temp(2)), (T.AGG_AVG, T.col(2)), (T.AGG_SUM, T.col(3)),
            (T.AGG_COUNT_STAR, None), (T.AGG_MAX, T.col(3))]
    kw = dict(instrs=instrs, consts=[3.0], aggs=aggs)
    yield "compact 2 keys + predicate", T.make_agg_config(T.AGG_COMPACT_KEY, layout, keys=[0, 1], est_groups=15000,
                                                          pred=[(2, T.GE, -900), (5, T.LT, 100000.0)], **kw), cols
    yield "generic 15 k groups (LDS overflow / hash ranges)", T.make_agg_config(T.AGG_GENERIC, layout, keys=[1, 0], est_groups=15000, **kw), cols
    yield "single state", T.make_agg_config(T.AGG_SINGLE_STATE, layout, **kw), cols
    skey = np.sort(rng.integers(0, 5000, size=n)).astype(np.int32)
    yield "collision free", T.make_agg_config(T.AGG_COLLISION_FREE, layout, keys=[0], num_entries=5000, **kw), [skey] + cols[1:]
    yield "q1 with date predicate", q1_config(with_date_pred=True), q1_columns(rng, n, with_date=True)


@pytest.mark.parametrize("with_filter", [False, True])
def test_run_time_plan_shapes_match_interpreter_and_oracle(capi, oracle, dev, jit_now, monkeypatch, with_filter):
    rng = np.random.default_rng(77)
    n = 400_003
    for name, cfg, cols in configs(rng, n):
        f = oracle.bitmap_from_bools(rng.random(n) < 0.6) if with_filter else None
        jit = run_hip(capi, dev, cfg, cols, filter_bitmap=f)
        got = finalize_np(jit, dev)
        monkeypatch.setenv("QSX_AGG_JIT_MIN_ROWS", str(1 << 60))           # this state: interpreter only
        interp = finalize_np(run_hip(capi, dev, cfg, cols, filter_bitmap=f), dev)
        monkeypatch.setenv("QSX_AGG_JIT_MIN_ROWS", "0")
        o = oracle.AggState(cfg)
        o.update(cols, filter_bitmap=f)
        ref = o.finalize()
        assert_same_groups(got, ref), name
        assert_same_groups(got, interp), name
        # two more blocks through the (now cached) specialised kernel + a merge of partial states
        again = run_hip(capi, dev, cfg, [c[: n // 2] for c in cols],
                        filter_bitmap=None if f is None else oracle.bitmap_from_bools(oracle.bools_from_bitmap(f, n)[: n // 2]))
        rest = run_hip(capi, dev, cfg, [c[n // 2:] for c in cols],
                       filter_bitmap=None if f is None else oracle.bitmap_from_bools(oracle.bools_from_bitmap(f, n)[n // 2:]))
        again.import_merge(rest.export(dev))
        assert_same_groups(finalize_np(again, dev), ref), name


def test_jit_disabled_by_environment_still_correct(capi, oracle, dev, monkeypatch):
    monkeypatch.setenv("QSX_AGG_JIT_MIN_ROWS", "0")
    monkeypatch.setenv("QSX_AGG_NO_SPECIALIZE", "1")
    rng = np.random.default_rng(5)
    name, cfg, cols = next(configs(rng, 50_000))
    o = oracle.AggState(cfg)
    o.update(cols)
    assert_same_groups(finalize_np(run_hip(capi, dev, cfg, cols), dev), o.finalize())


def test_background_compile_does_not_stall_and_takes_over(capi, oracle, dev, monkeypatch):
    """With a non-zero row threshold the hipRTC compile runs in a background thread: the update that triggers it returns on
    the interpreter (state 0 = compiling or already 1), later updates pick the shape up (state 1), and the groups are
    the oracle's whichever kernel served which block."""
    import ctypes
    import time
    monkeypatch.setenv("QSX_AGG_JIT_MIN_ROWS", "1000")
    monkeypatch.delenv("QSX_AGG_JIT_SYNC", raising=False)
    monkeypatch.delenv("QSX_AGG_NO_SPECIALIZE", raising=False)
    rng = np.random.default_rng(123)
    n = 50_000
    key = rng.integers(0, 37, size=n).astype(np.int32)
    val = rng.normal(size=n)
    odd = rng.integers(0, 1000, size=n).astype(np.int64)
    # a configuration nothing else in the suite compiles (the cache is per process and keyed by source text)
    cfg = T.make_agg_config(T.AGG_COMPACT_KEY, [(T.INT, None), (T.DOUBLE, None), (T.LONG, None)], keys=[0],
                            instrs=[(T.EX_MUL, 0, T.col(1), T.const(0))], consts=[1.0009765625],
                            aggs=[(T.AGG_SUM, T.temp(0)), (T.AGG_MAX, T.col(2)), (T.AGG_COUNT_STAR, None)], est_groups=64)
    state_of = capi.lib.qsx_debug_agg_jit_state
    state_of.restype = ctypes.c_int
    st = capi.AggState(cfg)
    o = oracle.AggState(cfg)
    cols = [to_dev(key, dev), to_dev(val, dev), to_dev(odd, dev)]
    assert state_of(st._h, 0) == -2                                   # nothing requested yet
    t0 = time.perf_counter()
    st.update(cols, n)
    first_call = time.perf_counter() - t0
    o.update([key, val, odd])
    assert state_of(st._h, 0) in (0, 1)
    blocks = 1
    deadline = time.time() + 60
    while state_of(st._h, 0) == 0 and time.time() < deadline:
        time.sleep(0.05)
        st.update(cols, n)                                            # served by the interpreter meanwhile
        o.update([key, val, odd])
        blocks += 1
    assert state_of(st._h, 0) == 1, "the run-time plan shape never became ready"
    st.update(cols, n)                                                # served by the compiled shape
    o.update([key, val, odd])
    assert_same_groups(finalize_np(st, dev), o.finalize())
    assert first_call < 0.5 or blocks == 1, f"the triggering update stalled for {first_call:.2f} s"


_CACHE_SCRIPT = r"""
import ctypes, json, sys, time
import numpy as np, torch
import quickstep_amd.capi as capi
from quickstep_amd import types as T
dev = torch.device("cuda:0")
rng = np.random.default_rng(7)
n = 200_000
key = rng.integers(0, 23, size=n).astype(np.int32)
val = rng.integers(0, 1000, size=n).astype(np.float64)
cfg = T.make_agg_config(T.AGG_COMPACT_KEY, [(T.INT, None), (T.DOUBLE, None)], keys=[0],
                        instrs=[(T.EX_MUL, 0, T.col(1), T.const(0))], consts=[1.5],
                        aggs=[(T.AGG_SUM, T.temp(0)), (T.AGG_COUNT_STAR, None)], est_groups=64)
state_of = capi.lib.qsx_debug_agg_jit_state
state_of.restype = ctypes.c_int
st = capi.AggState(cfg)
cols = [torch.from_numpy(key).to(dev), torch.from_numpy(val).to(dev)]
st.update(cols, n)
after_first = state_of(st._h, 0)
deadline = time.time() + 60
while state_of(st._h, 0) == 0 and time.time() < deadline:
    time.sleep(0.05)
st.update(cols, n)
keys, vals, nulls, groups = st.finalize(dev, capacity=64)
g = int(groups.item())
order = torch.argsort(keys[0][:g])
print(json.dumps({"after_first_update": after_first, "final": state_of(st._h, 0),
                  "sums": [float(x) for x in vals[0][:g][order].cpu()], "keys": [int(x) for x in keys[0][:g][order].cpu()]}))
"""


def test_plan_shape_from_the_cache_directory_serves_the_first_update(tmp_path):
    """QSX_JIT_CACHE_DIR across processes: the first process compiles the shape in the background (its first update runs on the
    interpreter) and leaves the code object on disk; the second process loads it before its first update returns.  Both
    give the same sums (integer-valued doubles: exact in any order)."""
    import json
    import os
    import subprocess
    import sys
    # (QSX_JIT_SHIPPED_CACHE=0: this shape is among the recorded ones — the code object that ships with the library would
    # answer the first process before it compiled or wrote anything)
    env = dict(os.environ, QSX_JIT_CACHE_DIR=str(tmp_path), QSX_AGG_JIT_MIN_ROWS="1000", QSX_JIT_SHIPPED_CACHE="0")
    env.pop("QSX_AGG_JIT_SYNC", None)
    env.pop("QSX_AGG_NO_SPECIALIZE", None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    runs = []
    for _ in range(2):
        r = subprocess.run([sys.executable, "-c", _CACHE_SCRIPT], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        runs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert runs[0]["final"] == 1 and runs[1]["final"] == 1
    assert runs[0]["after_first_update"] in (0, 1)
    assert runs[1]["after_first_update"] == 1, "the cached code object was not picked up before the first update returned"
    assert len([p for p in tmp_path.iterdir() if p.suffix == ".hsaco"]) >= 1
    assert runs[0]["keys"] == runs[1]["keys"] == list(range(23)) and runs[0]["sums"] == runs[1]["sums"]


def test_a_recorded_plan_shape_is_served_by_the_shipped_code_object(tmp_path):
    """quickstep_amd/lib/jit_cache (filled by __graft_entry__.build from csrc/jit_shapes/): the FIRST process that asks for a
    recorded shape gets it before its first update returns — no compiler, no interpreter, nothing written — where the
    same process with QSX_JIT_SHIPPED_CACHE=0 starts on the interpreter and compiles."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shipped = os.path.join(root, "quickstep_amd", "lib", "jit_cache")
    if not os.path.isdir(shipped) or not any(f.endswith(".hsaco") for f in os.listdir(shipped)):
        pytest.skip("no shipped code objects (build() has not run its warm step)")
    env = dict(os.environ, QSX_AGG_JIT_MIN_ROWS="1000")
    for name in ("QSX_JIT_CACHE_DIR", "QSX_AGG_JIT_SYNC", "QSX_AGG_NO_SPECIALIZE", "QSX_JIT_SHIPPED_CACHE", "QSX_JIT_RECORD_DIR"):
        env.pop(name, None)
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    before = sorted(os.listdir(shipped))
    r = subprocess.run([sys.executable, "-c", _CACHE_SCRIPT], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    run = json.loads(r.stdout.strip().splitlines()[-1])
    assert run["after_first_update"] == 1 and run["final"] == 1, "the shipped code object did not serve the first update (re-record csrc/jit_shapes?)"
    assert sorted(os.listdir(shipped)) == before
    assert run["keys"] == list(range(23))
