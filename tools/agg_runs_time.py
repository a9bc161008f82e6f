import sys, os, json, torch
sys.path.insert(0, "/root/repo")
import quickstep_amd.capi as capi, bench
dev = torch.device("cuda:0")
n = 600_000_000
cols = bench.gen_q1_columns_gpu(n, dev, 4)
st = capi.AggState(bench.q1_config())
for block in (n, 4_000_000, 120_000):
    blocks = [[c[s:min(n, s + block)] for c in cols] for s in range(0, n, block)]
    def run():
        st.clear(); st.update_blocks(blocks)
    run(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); run(); b.record(); torch.cuda.synchronize()
    print(json.dumps({"rows": n, "block_rows": block, "blocks": len(blocks), "one_run_ms": round(a.elapsed_time(b) / 2, 3)}), flush=True)
