"""CPU: the C-ABI library loads and exports every symbol include/qsx.h declares
(no compute calls: there is no GPU here), and refuses to compute without one."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "qsx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qsx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(capi):
    names = declared_functions()
    assert len(names) >= 40
    lib = ctypes.CDLL(capi.LIB_PATH)
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/qsx.h but not exported by libqsx.so"
    # and the Python binding table covers exactly the declared functions
    assert sorted(capi.EXPORTED) == names


def test_abi_version_and_struct_mirror(capi):
    from quickstep_amd import types as T
    assert capi.lib.qsx_abi_version() == T.ABI_VERSION == 19
    header = open(os.path.join(ROOT, "include", "qsx.h")).read()
    assert f"#define QSX_ABI_VERSION {T.ABI_VERSION}" in header
    assert capi.lib.qsx_abi_sizeof_agg_config() == ctypes.sizeof(T.AggConfig)
    assert capi.lib.qsx_status_string(0) == b"ok"
    assert b"no CPU" in capi.lib.qsx_status_string(T.ERR_NO_DEVICE)


def test_no_cpu_fallback_without_a_gpu(capi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from quickstep_amd import types as T
    assert capi.device_count() == 0
    h = ctypes.c_void_p()
    assert capi.lib.qsx_join_table_create(T.INT, 16, ctypes.byref(h)) == T.ERR_NO_DEVICE
    buf = np.zeros(4, dtype=np.int32)
    lit = ctypes.c_int32(1)
    rc = capi.lib.qsx_select_cmp(T.INT, buf.ctypes.data, 4, T.LT, ctypes.byref(lit), None, buf.ctypes.data, None, None)
    assert rc == T.ERR_NO_DEVICE
    cfg = T.make_agg_config(T.AGG_SINGLE_STATE, [(T.INT, None)], aggs=[(T.AGG_COUNT_STAR, None)])
    assert capi.lib.qsx_agg_state_create(ctypes.byref(cfg), ctypes.byref(h)) == T.ERR_NO_DEVICE


def test_product_code_never_touches_the_oracle():
    """quickstep_amd/ (the product) must not import, include or link oracle/."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "quickstep_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", "Makefile")):
                text = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"^\s*(import|from)\s+\S*oracle|#include\s*[\"<][^\">]*oracle|qso_|libqsx_oracle|pyoracle\s*\.",
                             text, flags=re.M):
                    bad.append(os.path.join(base, f))
    assert bad == []


def test_run_time_plan_shape_generator_compiles_without_a_gpu(capi):
    """csrc/agg_jit.hip: the source generator + hipRTC build of a plan shape (gfx950 code object) needs no
    device, so the CPU suite covers it: hash, dense and filter variants, MIN/MAX, predicates."""
    import ctypes as C
    from quickstep_amd import types as T
    fn = capi.lib.qsx_debug_jit_compile
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(T.AggConfig), C.c_int, C.POINTER(C.c_size_t)]
    layout = [(T.INT, None), (T.LONG, None), (T.FLOAT, None), (T.DOUBLE, None)]
    kw = dict(instrs=[(T.EX_MUL, 0, T.col(3), T.col(2)), (T.EX_DIV, 1, T.temp(0), T.const(0))], consts=[-0.75],
              aggs=[(T.AGG_SUM, T.temp(1)), (T.AGG_MIN, T.col(2)), (T.AGG_MAX, T.col(1)), (T.AGG_AVG, T.col(0)), (T.AGG_COUNT_STAR, None)],
              pred=[(1, T.GT, -5), (3, T.LE, 2.5)])
    # (bit 1 of the second argument: the group-directory variant of a hash strategy)
    for cfg, with_filter in ((T.make_agg_config(T.AGG_COMPACT_KEY, layout, keys=[0], **kw), 0),
                             (T.make_agg_config(T.AGG_COLLISION_FREE, layout, keys=[0], num_entries=100, **kw), 1),
                             (T.make_agg_config(T.AGG_GENERIC, layout, keys=[0, 2], **kw), 2),
                             (T.make_agg_config(T.AGG_COMPACT_KEY, layout, keys=[0], **kw), 3)):
        size = C.c_size_t(0)
        assert fn(C.byref(cfg), with_filter, C.byref(size)) == 0
        assert size.value > 1000


def test_plan_shapes_over_nullable_columns_compile(capi, tmp_path, monkeypatch):
    """A state over nullable columns gets a run-time plan shape too (round 4: the interpreter served them before): the null
    slots, their LDS offsets and the accumulators' null masks are constants of the shape, the bitmaps of a call arrive
    behind the `nulls` pointer — plain, filtered, group-directory and dense flavours."""
    import ctypes as C
    from quickstep_amd import types as T
    fn = capi.lib.qsx_debug_jit_compile
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(T.AggConfig), C.c_int, C.POINTER(C.c_size_t)]
    layout = [(T.INT, None), (T.DOUBLE, None), (T.LONG, None)]
    aggs = [(T.AGG_SUM, T.col(1)), (T.AGG_COUNT, T.col(2)), (T.AGG_AVG, T.col(1)), (T.AGG_MIN, T.col(2)), (T.AGG_COUNT_STAR, None)]
    for strategy, extra, bits in ((T.AGG_GENERIC, {}, 0), (T.AGG_GENERIC, {}, 1), (T.AGG_GENERIC, {}, 2),
                                  (T.AGG_COLLISION_FREE, {"num_entries": 5000}, 0), (T.AGG_COLLISION_FREE, {"num_entries": 5000}, 8)):
        cfg = T.make_agg_config(strategy, layout, keys=[0], aggs=aggs, nullable=[0, 1, 2], **extra)
        dump = tmp_path / f"nullable_{strategy}_{bits}.hip"
        monkeypatch.setenv("QSX_JIT_DUMP", str(dump))
        size = C.c_size_t(0)
        assert fn(C.byref(cfg), bits, C.byref(size)) == 0 and size.value > 1000
        text = dump.read_text()
        assert "d.num_null_cols = 3" in text and "const unsigned long long *const *nulls" in text and "count_valid = 1" in text


def test_plan_shape_flavours_have_sources_of_their_own(capi, tmp_path, monkeypatch):
    """The run-of-blocks flavour and the dense-state-in-LDS flavour of a plan shape are generated as such (template
    arguments emitted, not patched into the text): every flavour compiles and its translation unit differs from the plain
    one's — a run flavour that silently came out as the single-stripe kernel would read past the first block."""
    import ctypes as C
    from quickstep_amd import types as T
    fn = capi.lib.qsx_debug_jit_compile
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(T.AggConfig), C.c_int, C.POINTER(C.c_size_t)]
    layout = [(T.INT, None), (T.DOUBLE, None)]
    aggs = [(T.AGG_SUM, T.col(1)), (T.AGG_COUNT_STAR, None)]
    sources = {}
    for name, cfg, bits in (("hash", T.make_agg_config(T.AGG_COMPACT_KEY, layout, keys=[0], aggs=aggs), 0),
                            ("hash_runs", T.make_agg_config(T.AGG_COMPACT_KEY, layout, keys=[0], aggs=aggs), 4),
                            ("dense", T.make_agg_config(T.AGG_COLLISION_FREE, layout, keys=[0], aggs=aggs, num_entries=5000), 0),
                            ("dense_runs", T.make_agg_config(T.AGG_COLLISION_FREE, layout, keys=[0], aggs=aggs, num_entries=5000), 4),
                            ("dense_lds", T.make_agg_config(T.AGG_COLLISION_FREE, layout, keys=[0], aggs=aggs, num_entries=5000), 8),
                            ("dense_lds_runs", T.make_agg_config(T.AGG_COLLISION_FREE, layout, keys=[0], aggs=aggs, num_entries=5000), 12)):
        dump = tmp_path / f"{name}.hip"
        monkeypatch.setenv("QSX_JIT_DUMP", str(dump))
        size = C.c_size_t(0)
        assert fn(C.byref(cfg), bits, C.byref(size)) == 0 and size.value > 1000
        sources[name] = dump.read_text()
    assert len(set(sources.values())) == len(sources)
    tail = lambda text: text[text.rindex("agg_hash_update_body<"):]   # noqa: E731
    assert "false, true>(D, nullptr" in tail(sources["hash_runs"]) and "false, true>(D, nullptr" in tail(sources["dense_runs"])
    assert "true>(D, nullptr" in tail(sources["dense_lds_runs"]) and "(D, cols.p" in tail(sources["dense_lds"])
    assert "(D, cols.p" in tail(sources["hash"]) and "(D, cols.p" in tail(sources["dense"])


def test_plan_shape_code_objects_are_kept_in_the_cache_directory(capi, tmp_path, monkeypatch):
    """QSX_JIT_CACHE_DIR: the first build of a plan shape leaves one file there (source text + code object), the second
    takes the code object from it (same bytes, no compile); a damaged or foreign file is ignored and replaced."""
    import ctypes as C
    import time
    from quickstep_amd import types as T
    fn = capi.lib.qsx_debug_jit_compile
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(T.AggConfig), C.c_int, C.POINTER(C.c_size_t)]
    layout = [(T.INT, None), (T.DOUBLE, None)]
    cfg = T.make_agg_config(T.AGG_COMPACT_KEY, layout, keys=[0], aggs=[(T.AGG_SUM, T.col(1)), (T.AGG_COUNT_STAR, None)])
    monkeypatch.setenv("QSX_JIT_CACHE_DIR", str(tmp_path))
    monkeypatch.setenv("QSX_JIT_SHIPPED_CACHE", "0")               # (the code objects that ship with the library would answer first)
    size = C.c_size_t(0)
    t0 = time.perf_counter()
    assert fn(C.byref(cfg), 0, C.byref(size)) == 0
    compile_s = time.perf_counter() - t0
    files = sorted(p for p in tmp_path.iterdir())
    assert len(files) == 1 and files[0].name.startswith("qsx_") and files[0].suffix == ".hsaco"
    blob = files[0].read_bytes()
    assert blob[:8] == b"QSXJIT01" and b"qsx_jit_agg" in blob and len(blob) > size.value
    first = size.value
    t0 = time.perf_counter()
    assert fn(C.byref(cfg), 0, C.byref(size)) == 0
    cached_s = time.perf_counter() - t0
    assert size.value == first and cached_s < compile_s / 3, (compile_s, cached_s)
    assert sorted(p for p in tmp_path.iterdir()) == files
    # another plan shape: another file
    other = T.make_agg_config(T.AGG_COMPACT_KEY, layout, keys=[0], aggs=[(T.AGG_MIN, T.col(1))])
    assert fn(C.byref(other), 0, C.byref(size)) == 0
    assert len(list(tmp_path.iterdir())) == 2
    # a truncated file is not trusted: recompiled and rewritten whole
    files[0].write_bytes(blob[:len(blob) // 2])
    assert fn(C.byref(cfg), 0, C.byref(size)) == 0 and size.value == first
    assert files[0].read_bytes() == blob


def test_plan_shapes_name_their_compiler_in_the_cache(capi, tmp_path, monkeypatch):
    """Run-time plan shapes go to the ROCm compiler driver (a child process) when one is installed and to hipRTC otherwise
    (`QSX_JIT_COMPILER`): objects of the two never share a cache file, and a driver that cannot be run
    leaves the shape to hipRTC instead of failing the state."""
    import ctypes as C
    from quickstep_amd import types as T
    fn = capi.lib.qsx_debug_jit_compile
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(T.AggConfig), C.c_int, C.POINTER(C.c_size_t)]
    layout = [(T.INT, None), (T.DOUBLE, None)]
    cfg = T.make_agg_config(T.AGG_COMPACT_KEY, layout, keys=[0], aggs=[(T.AGG_MAX, T.col(1))])
    monkeypatch.setenv("QSX_JIT_CACHE_DIR", str(tmp_path))
    monkeypatch.setenv("QSX_JIT_SHIPPED_CACHE", "0")
    size = C.c_size_t(0)
    stamps = {}
    for compiler in ("hiprtc", None, str(tmp_path / "no_such_driver")):
        if compiler is None:
            monkeypatch.delenv("QSX_JIT_COMPILER", raising=False)
        else:
            monkeypatch.setenv("QSX_JIT_COMPILER", compiler)
        before = set(tmp_path.iterdir())
        assert fn(C.byref(cfg), 0, C.byref(size)) == 0 and size.value > 0
        new = set(tmp_path.iterdir()) - before
        if compiler is not None and compiler != "hiprtc":
            assert not new                      # no driver there: the same key as hipRTC's, already in the directory
            continue
        assert len(new) == 1
        blob = new.pop().read_bytes()
        stamps[compiler] = blob[24:blob.index(b"\n", 24)]
    assert stamps["hiprtc"].startswith(b"hiprtc ")
    if os.access(os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "bin", "hipcc"), os.X_OK):
        assert stamps[None].startswith(b"driver ") and b"hipcc" in stamps[None]


def test_device_selection_needs_a_device(capi):
    """qsx_current_device / qsx_set_current_device (include/qsx.h): the calling thread's HIP device, for engines that pick
    their GPU per thread instead of per process; without a GPU both say so."""
    import ctypes as C
    from quickstep_amd import types as T
    if capi.lib.qsx_device_count() > 0:
        pytest.skip("a GPU is present")
    device = C.c_int(-7)
    assert capi.lib.qsx_current_device(C.byref(device)) == T.ERR_NO_DEVICE
    assert capi.lib.qsx_set_current_device(0) == T.ERR_NO_DEVICE


def test_plan_shapes_over_code_stripes_use_no_scratch(capi, tmp_path, monkeypatch):
    """Plan shapes keep their state in registers and LDS.  The values of compressed attributes decoded into registers
    (DecodedRows) once sat in 520 bytes of scratch per lane — a null test on the struct's address kept it in memory — and the
    hipRTC build of that shape produced wrong sums: every flavour of a shape over code stripes (a narrow decoded column in
    the predicate and in an integer SUM included) must come out of both compilers without a private segment."""
    import ctypes as C
    import shutil
    import subprocess
    from quickstep_amd import types as T
    readelf, bundler = "/opt/rocm/lib/llvm/bin/llvm-readelf", "/opt/rocm/lib/llvm/bin/clang-offload-bundler"
    if not (os.path.exists(readelf) and os.path.exists(bundler)):
        pytest.skip("no llvm-readelf / clang-offload-bundler in this image")
    fn = capi.lib.qsx_debug_jit_compile
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(T.AggConfig), C.c_int, C.POINTER(C.c_size_t)]
    layout = [(T.CHAR, 1), (T.CHAR, 1), (T.DOUBLE, None), (T.DOUBLE, None), (T.DOUBLE, None), (T.DOUBLE, None), (T.INT, None)]
    cfg = T.make_agg_config(T.AGG_COMPACT_KEY, layout, code_widths=[0, 0, 1, 0, 1, 1, 1], keys=[0, 1],
                            instrs=[(T.EX_SUB, 0, T.const(0), T.col(4)), (T.EX_MUL, 1, T.col(3), T.temp(0))], consts=[1.0],
                            aggs=[(T.AGG_SUM, T.col(2)), (T.AGG_SUM, T.temp(1)), (T.AGG_COUNT_STAR, None), (T.AGG_SUM, T.col(6)),
                                  (T.AGG_MAX, T.col(5))], pred=[(6, T.LT, 7)], est_groups=6)
    # (and the dense per-row path reading everything through a pair list: 4-byte "codes" = row numbers)
    dense = T.make_agg_config(T.AGG_COLLISION_FREE, [(T.INT, None), (T.DOUBLE, None), (T.DOUBLE, None)], keys=[0],
                              instrs=[(T.EX_SUB, 0, T.const(0), T.col(2)), (T.EX_MUL, 1, T.col(1), T.temp(0))], consts=[1.0],
                              aggs=[(T.AGG_SUM, T.temp(1)), (T.AGG_COUNT_STAR, None)], num_entries=60_000_000, code_widths=[4, 4, 4])
    for compiler in ("", "hiprtc"):
        if compiler:
            monkeypatch.setenv("QSX_JIT_COMPILER", compiler)
        else:
            monkeypatch.delenv("QSX_JIT_COMPILER", raising=False)
        for bits in (0, 1, 4, 16, 32, 33):          # plain, filtered, run of blocks, register groups; dense plain, filtered
            if bits >= 32:
                cfg_used, bits = dense, bits - 32
            else:
                cfg_used = cfg
            code = tmp_path / f"shape_{compiler or 'driver'}_{bits}.co"
            monkeypatch.setenv("QSX_JIT_DUMP_CODE", str(code))
            size = C.c_size_t(0)
            assert fn(C.byref(cfg_used), bits, C.byref(size)) == 0 and size.value > 1000
            elf = tmp_path / "shape.elf"
            r = subprocess.run([bundler, "--unbundle", "--type=o", f"--input={code}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                f"--output={elf}"], capture_output=True)
            if r.returncode != 0:           # (hipRTC hands out the bare code object)
                shutil.copy(code, elf)
            notes = subprocess.run([readelf, "--notes", str(elf)], capture_output=True, text=True).stdout
            assert ".private_segment_fixed_size: 0" in notes, (compiler, bits, [ln for ln in notes.splitlines() if "private_segment" in ln])


def test_recorded_plan_shapes_are_compiled_at_build_time_and_found_at_run_time(capi, tmp_path, monkeypatch):
    """The shipped code objects (csrc/agg_jit.hip: QSX_JIT_RECORD_DIR -> csrc/jit_shapes/*.shape -> qsx_jit_warm at build time
    -> <library directory>/jit_cache at run time): a process that asks for a plan shape leaves its text in the record
    directory; qsx_jit_warm compiles that text into a directory under the name the run-time lookup computes (a second call
    finds it there: 1); and the library directory's jit_cache answers a later request without a compile and without writing
    anything.  No GPU involved: hipcc cross-compiles."""
    import ctypes as C
    import os
    import shutil
    import time
    from quickstep_amd import types as T
    fn = capi.lib.qsx_debug_jit_compile
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(T.AggConfig), C.c_int, C.POINTER(C.c_size_t)]
    warm = capi.lib.qsx_jit_warm
    warm.restype = C.c_int
    warm.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
    if not os.access("/opt/rocm/bin/hipcc", os.X_OK):
        pytest.skip("no compiler driver: nothing is shipped from a hipRTC-only installation")
    layout = [(T.INT, None), (T.LONG, None), (T.DOUBLE, None)]
    cfg = T.make_agg_config(T.AGG_COMPACT_KEY, layout, keys=[0, 1], aggs=[(T.AGG_SUM, T.col(2)), (T.AGG_MAX, T.col(2)), (T.AGG_COUNT_STAR, None)])
    record, built = tmp_path / "record", tmp_path / "built"
    record.mkdir()
    built.mkdir()
    monkeypatch.setenv("QSX_JIT_RECORD_DIR", str(record))
    monkeypatch.setenv("QSX_JIT_SHIPPED_CACHE", "0")
    monkeypatch.delenv("QSX_JIT_CACHE_DIR", raising=False)
    size = C.c_size_t(0)
    assert fn(C.byref(cfg), 0, C.byref(size)) == 0
    shapes = sorted(record.iterdir())
    assert len(shapes) == 1 and shapes[0].suffix == ".shape"
    text = shapes[0].read_bytes()
    assert b"jit_make_dev" in text and b"qsx_jit_agg" in text and len(text) < 64 * 1024      # the configuration and the entry point, not the kernel sources
    assert fn(C.byref(cfg), 0, C.byref(size)) == 0 and sorted(record.iterdir()) == shapes      # recorded once
    monkeypatch.delenv("QSX_JIT_RECORD_DIR")
    assert warm(text, len(text), str(built).encode()) == 0
    objects = sorted(built.iterdir())
    assert len(objects) == 1 and objects[0].suffix == ".hsaco" and objects[0].read_bytes()[:8] == b"QSXJIT01"
    assert warm(text, len(text), str(built).encode()) == 1                                      # already there
    # the run-time side: the same object under <library directory>/jit_cache is found (no compile: an order of magnitude faster)
    shipped = os.path.join(os.path.dirname(capi.LIB_PATH), "jit_cache")
    made = not os.path.isdir(shipped)
    os.makedirs(shipped, exist_ok=True)
    target = os.path.join(shipped, objects[0].name)
    present = os.path.exists(target)
    try:
        if not present:
            shutil.copy(objects[0], target)
        monkeypatch.setenv("QSX_JIT_SHIPPED_CACHE", "1")
        before = sorted(os.listdir(shipped))
        t0 = time.perf_counter()
        assert fn(C.byref(cfg), 0, C.byref(size)) == 0 and size.value > 0
        assert time.perf_counter() - t0 < 0.5, "the shipped code object was not used"
        assert sorted(os.listdir(shipped)) == before                                            # nothing is written next to the library
    finally:
        if not present:
            os.remove(target)
        if made:
            shutil.rmtree(shipped, ignore_errors=True)


def test_which_plans_factor_through_their_dictionary_columns(capi):
    """csrc/agg_factored.hpp factored_analyse (host logic, no GPU): an aggregation over dictionary-coded attributes factors when
    every aggregate argument is affine in the plain columns once the dictionary columns are fixed.  TPC-H Q1 over lineitem's
    codes: cells over (discount, tax), a histogram for quantity, one carrier (extendedprice)."""
    import ctypes as C
    from quickstep_amd import types as T
    fn = capi.lib.qsx_debug_agg_factored_plan
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(T.AggConfig), C.POINTER(C.c_int32), C.c_int]

    def plan(cfg):
        out = (C.c_int32 * 16)()
        assert fn(C.byref(cfg), out, 16) == 0
        return list(out)
    layout = [(T.CHAR, 1), (T.CHAR, 1), (T.DOUBLE, None), (T.DOUBLE, None), (T.DOUBLE, None), (T.DOUBLE, None)]
    q1 = dict(keys=[0, 1], instrs=[(T.EX_SUB, 0, T.const(0), T.col(4)), (T.EX_MUL, 1, T.col(3), T.temp(0)), (T.EX_ADD, 2, T.const(0), T.col(5)),
                                   (T.EX_MUL, 3, T.temp(1), T.temp(2))], consts=[1.0],
              aggs=[(T.AGG_SUM, T.col(2)), (T.AGG_SUM, T.col(3)), (T.AGG_SUM, T.temp(1)), (T.AGG_SUM, T.temp(3)), (T.AGG_AVG, T.col(4)), (T.AGG_COUNT_STAR, None)],
              est_groups=6)
    coded = [0, 0, 1, 0, 1, 1]
    got = plan(T.make_agg_config(T.AGG_COMPACT_KEY, layout, code_widths=coded, **q1))
    assert got[:4] == [1, 2, 1, 1]                     # factors; cells (disc, tax); histogram (qty); carrier (price)
    assert got[4] == 0 and got[5:9] == [-1, -1, -1, -1]   # SUM(qty) reads the histogram, the others the cells
    # the same plan over plain columns has nothing to factor through
    assert plan(T.make_agg_config(T.AGG_COMPACT_KEY, layout, **q1))[0] == 0
    # price * price is not affine; MIN / MAX do not factor; a predicate inside the state does not stand in the way (a pass of
    # its own turns it into the call's filter)
    square = dict(q1, instrs=[(T.EX_MUL, 0, T.col(3), T.col(3)), (T.EX_MUL, 1, T.temp(0), T.col(4))], aggs=[(T.AGG_SUM, T.temp(1))])
    assert plan(T.make_agg_config(T.AGG_COMPACT_KEY, layout, code_widths=coded, **square))[0] == 0
    minmax = dict(q1, aggs=[(T.AGG_SUM, T.col(4)), (T.AGG_MAX, T.col(3))])
    assert plan(T.make_agg_config(T.AGG_COMPACT_KEY, layout, code_widths=coded, **minmax))[0] == 0
    assert plan(T.make_agg_config(T.AGG_COMPACT_KEY, layout, code_widths=coded, pred=[(3, T.LT, 1000.0)], **q1))[:4] == [1, 2, 1, 1]
    # price + disc is affine (the constant part multiplies the cell's count); price / (1 + tax) divides by a dictionary-only term
    affine = dict(q1, instrs=[(T.EX_ADD, 0, T.col(3), T.col(4)), (T.EX_ADD, 1, T.const(0), T.col(5)), (T.EX_DIV, 2, T.col(3), T.temp(1))],
                  aggs=[(T.AGG_SUM, T.temp(0)), (T.AGG_SUM, T.temp(2))])
    assert plan(T.make_agg_config(T.AGG_COMPACT_KEY, layout, code_widths=coded, **affine))[:4] == [1, 2, 0, 1]
    # ... a dictionary column in the numerator's place is fine, a plain column in the denominator is not
    bad_div = dict(q1, instrs=[(T.EX_DIV, 0, T.col(4), T.col(3))], aggs=[(T.AGG_SUM, T.temp(0))])
    assert plan(T.make_agg_config(T.AGG_COMPACT_KEY, layout, code_widths=coded, **bad_div))[0] == 0
