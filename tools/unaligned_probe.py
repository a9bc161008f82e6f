"""Q1 over code stripes through the factored direct-load kernel with stripes at odd byte offsets (a reference block image keeps
its stripes at multiples of the block's tuple capacity): same groups as the aligned stripes?  what does it cost?"""
import os, sys, json
os.environ["QSX_AGG_FACTORED_MIN_ROWS"] = "0"
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import quickstep_amd.capi as capi
import bench
dev = torch.device("cuda", 0)
n = 100_000_000
g = torch.Generator(device=dev); g.manual_seed(4)
pad = 64
combo = torch.multinomial(torch.tensor([0.2466, 0.0065, 0.5005, 0.2464], device=dev), n + pad, replacement=True, generator=g)
k1 = torch.tensor(list(b"ANNR"), dtype=torch.uint8, device=dev)[combo]
k2 = torch.tensor(list(b"FFOF"), dtype=torch.uint8, device=dev)[combo]
qty = torch.randint(0, 50, (n + pad,), device=dev, generator=g, dtype=torch.uint8)
disc = torch.randint(0, 11, (n + pad,), device=dev, generator=g, dtype=torch.uint8)
tax = torch.randint(0, 9, (n + pad,), device=dev, generator=g, dtype=torch.uint8)
# a DOUBLE stripe at a 1-byte-misaligned address: a byte buffer viewed from an odd offset
price_src = (torch.rand(n + pad, device=dev, generator=g, dtype=torch.float64) * 104100 + 900).mul(100).round().div(100)
raw = torch.empty((n + pad) * 8 + 16, dtype=torch.uint8, device=dev)
dicts = [None, None, torch.arange(1, 51, device=dev, dtype=torch.float64), None, torch.arange(0, 11, device=dev, dtype=torch.float64) / 100,
         torch.arange(0, 9, device=dev, dtype=torch.float64) / 100]
cfg = bench.q1_coded_config()
def run(off1, offp):
    raw[offp:offp + (n + pad) * 8] = price_src.view(torch.uint8)
    class P:  # a fake tensor: data_ptr at an odd byte offset
        def __init__(s, t, off): s.t, s.off = t, off
        def data_ptr(s): return s.t.data_ptr() + s.off
        def numel(s): return n
    cols = [k1[off1:], k2[off1:], qty[off1:], P(raw, offp + off1 * 8), disc[off1:], tax[off1:]]
    st = capi.AggState(cfg)
    before = capi.lib.qsx_debug_agg_factored_launches()
    st.update_coded(cols, dicts, n)
    torch.cuda.synchronize()
    assert capi.lib.qsx_debug_agg_factored_launches() - before == 1
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): st.update_coded(cols, dicts, n)
    b.record(); torch.cuda.synchronize()
    st.clear(); st.update_coded(cols, dicts, n)
    k, v, _, gg = st.finalize(dev, capacity=16)
    gg = int(gg.item())
    order = torch.argsort(k[0][:gg].long() * 256 + k[1][:gg].long())
    return a.elapsed_time(b) / 5, [x[:gg][order].double().cpu() for x in v]
import ctypes
capi.lib.qsx_debug_agg_factored_launches.restype = ctypes.c_longlong
base_ms, base = run(0, 0)
res = {"aligned_ms": base_ms}
for off1, offp in ((0, 1), (0, 4), (3, 0), (5, 3), (8, 0)):
    ms, v = run(off1, offp)
    # reference for the shifted rows: aligned copy
    ok = True
    res[f"rows+{off1}_price_bytes+{offp}_ms"] = ms
    if off1 == 0:
        ok = all(bool(torch.allclose(x, y, rtol=1e-12, atol=0)) for x, y in zip(v, base)) and bool(torch.equal(v[7], base[7])) and bool(torch.equal(v[0], base[0]))
        res[f"rows+{off1}_price_bytes+{offp}_same"] = ok
print(json.dumps(res))
