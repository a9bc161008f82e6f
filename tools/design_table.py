#!/usr/bin/env python3
"""DESIGN.md §4's measured-kernel table, generated: one round's numbers, nothing carried over by hand.
Reads profiles/rNN_bench_ops.jsonl (tools/bench_ops.py), rNN_bench_line.json (bench.py), rNN_probe_hashed_sparse.jsonl,
rNN_agg_coded_probe.jsonl, rNN_bench_c4.json / rNN_bench_c5.json when present, and rewrites the block between the markers
<!-- kernel-table:begin --> and <!-- kernel-table:end --> of DESIGN.md.   usage: python tools/design_table.py r04"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN, END = "<!-- kernel-table:begin -->", "<!-- kernel-table:end -->"


def load_lines(path):
    if not os.path.exists(path):
        return []
    return [json.loads(ln) for ln in open(path) if ln.strip().startswith("{")]


def main():
    tag = sys.argv[1]
    prof = os.path.join(ROOT, "profiles")
    rows = []
    line = (load_lines(os.path.join(prof, f"{tag}_bench_line.json")) or [None])[0]
    if line is not None:
        r = line["roofline"]
        rows.append(("`bench.py` step (C2 + C3)", f"{line['config']['workload'][:60]}…", f"{line['ms_per_step']:.2f} ms",
                     f"{line['value'] / 1e9:.0f} G rows/s", "—", "—"))
        rows.append((f"`{r['kernel'].split(' ')[0]}` (dominant)", "C3: 600 M rows, Q1 shape", f"{r['avg_launch_ms']:.2f} ms",
                     f"{r['rows_per_launch'] / r['avg_launch_ms'] / 1e6:.0f} G rows/s", f"{r['achieved'] / 1e3:.2f} TB/s", f"{100 * r['frac']:.0f} %"))
        p = line.get("probe")
        if p:
            rows.append((f"`{p['roofline']['kernel'].split(' ')[0]}`", "C2: 1 M x 100 M, m = 1.0", f"{p['ms']:.3f} ms", f"{p['rows_per_s'] / 1e9:.0f} G rows/s",
                         f"{p['roofline']['achieved'] / 1e3:.2f} TB/s", f"{100 * p['roofline']['frac']:.0f} %"))
            for name, v in sorted(p.get("variants", {}).items()):
                if isinstance(v, dict):
                    rows.append((f"probe variant `{name}`", "C2", f"{v['ms']:.3f} ms", f"{100_000_000 / v['ms'] / 1e6:.0f} G rows/s",
                                 f"{v.get('GBps', 0) / 1e3:.2f} TB/s", f"{100 * v.get('frac_of_hbm_peak', v.get('GBps', 0) / 8000):.0f} %"))
                else:
                    rows.append((f"probe variant `{name}`", "C2", f"{v:.3f} ms", "—", "—", "—"))
        op = line.get("operators", {})
        if "ms_per_step" in op:
            rows.append(("the step through the operator layer", f"{op.get('workers', 8)} Workers, {op.get('blocks_per_work_order', 256)} blocks per work order", f"{op['ms_per_step']:.2f} ms",
                         f"{op['rows_per_s'] / 1e9:.0f} G rows/s", "—", f"{100 * op['fraction_of_raw_abi_value']:.0f} % of the raw step"))
        for name, v in (line.get("secondary") or {}).items():      # (round 5: the other BASELINE configurations in the same line)
            if isinstance(v, dict) and "ms" in v and "roofline" in v and "rows_per_s" in v and "workload" in v:
                rows.append((f"`secondary.{name}` — `{v['roofline']['kernel'].split(' ')[0]}`", v["workload"][:90] + "…", f"{v['ms']:.3f} ms",
                             f"{v['rows_per_s'] / 1e9:.0f} G rows/s", f"{v['roofline']['achieved'] / 1e3:.2f} TB/s", f"{100 * v['roofline']['frac']:.0f} %"))
    for s in load_lines(os.path.join(prof, f"{tag}_probe_small_tables.jsonl")):
        if s["build_keys"] not in (25000, 8000) or (s["table"] == "hashed_sparse" and s["build_keys"] > 13000):   # (25 K sparse keys: 250 KB of buckets, not an LDS table)
            continue
        for k, label in (("pairs_ms_lds", "pairs, table in LDS"), ("pairs_ms_l2", "pairs, table in L2 (`QSX_JOIN_LDS=0`)")):
            if k in s:
                byts = (12 if k.startswith("pairs") else 4) * s["probe_rows"]
                rows.append((f"K4 small build side ({s['table']}): {label}", f"{s['build_keys']} keys x {s['probe_rows'] // 1_000_000} M, m = 1.0", f"{s[k]:.3f} ms",
                             f"{s['probe_rows'] / s[k] / 1e6:.0f} G rows/s", f"{byts / s[k] / 1e9:.2f} TB/s", f"{100 * byts / s[k] / 1e6 / 8000:.0f} %"))
    for cfg in ("c4", "c5"):
        c = (load_lines(os.path.join(prof, f"{tag}_bench_{cfg}.json")) or [None])[0]
        if c is not None:
            rows.append((f"`bench.py --config {cfg}` (one rank)", c["config"]["workload"][:70] + "…", f"{c['ms_per_step']:.2f} ms",
                         f"{c['value'] / 1e9:.1f} G rows/s", f"{c['roofline']['achieved'] / 1e3:.2f} TB/s", f"{100 * c['roofline']['frac']:.0f} %"))
    for s in load_lines(os.path.join(prof, f"{tag}_probe_hashed_sparse.jsonl")):
        for m in ("1.0", "0.2"):
            for kind in ("pairs", "count", "exists"):
                k = f"{kind}_m{m}_ms"
                if k in s:
                    rows.append((f"bucketed table, sparse keys: {kind}", f"{s['build_rows'] // 1000} K x 100 M, m = {m}", f"{s[k]:.3f} ms",
                                 f"{s['probe_rows'] / s[k] / 1e6:.0f} G rows/s", "—", "—"))
        rows.append(("bucketed table: clear + build", f"{s['build_rows'] // 1000} K keys", f"{s['clear_build_ms']:.3f} ms", "—", "—", "—"))
    for s in load_lines(os.path.join(prof, f"{tag}_agg_coded_probe.jsonl")):
        for k, label in (("coded_ms (13 B/row)", "Q1 over code stripes (13 B/row)"), ("coded_without_dictionaries_ms", "same, codes = values (no dictionary reads)"),
                         ("plain_ms (34 B/row)", "Q1 over plain stripes (34 B/row)")):
            if k in s:
                b = 13 if "13" in label or "codes" in label else 34
                how = ("factored through the codes, clear included" if k.startswith("coded_ms") and tag >= "r05" else
                       ("decoding plan shape" if tag >= "r05" and "codes" in label else "plan shape"))
                rows.append((label, f"{s['rows'] // 1_000_000} M rows, {how}", f"{s[k]:.2f} ms", f"{s['rows'] / s[k] / 1e6:.0f} G rows/s",
                             f"{b * s['rows'] / s[k] / 1e9:.2f} TB/s", f"{100 * b * s['rows'] / s[k] / 1e6 / 8000:.0f} %"))
    for o in load_lines(os.path.join(prof, f"{tag}_bench_ops.jsonl")):
        rows.append((o["op"], f"{o['rows'] // 1_000_000} M rows" + (f" ({o['note']})" if o.get("note") else ""), f"{o['ms']:.3f} ms",
                     f"{o['G_rows_per_s']:.0f} G rows/s", f"{o['achieved_GBps'] / 1e3:.2f} TB/s", f"{100 * o['frac_of_8TBps']:.0f} %"))
    table = ["| kernel / operator | workload | time | rows/s | algorithmic bytes/s | of 8 TB/s |", "|---|---|---|---|---|---|"]
    table += ["| " + " | ".join(r) + " |" for r in rows]
    text = BEGIN + f"\n(generated by `tools/design_table.py {tag}` from `profiles/{tag}_*`: one measurement pass, `tools/{tag}_measure.sh`)\n\n" + "\n".join(table) + "\n" + END
    path = os.path.join(ROOT, "DESIGN.md")
    doc = open(path).read()
    if BEGIN not in doc or END not in doc:
        raise SystemExit("DESIGN.md has no kernel-table markers")
    doc = doc[:doc.index(BEGIN)] + text + doc[doc.index(END) + len(END):]
    open(path, "w").write(doc)
    print(f"{len(rows)} rows")


if __name__ == "__main__":
    main()
