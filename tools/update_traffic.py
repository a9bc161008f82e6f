#!/usr/bin/env python3
"""profiles/traffic.json from a PMC summary (tools/prof_pmc.sh + tools/pmc_summary.py): HBM bytes per launch of the dominant
aggregation kernel = 2 x FETCH_SIZE (gfx950 reports half the bytes of a wide coalesced stream, MI355X_MICROARCH.md §HBM)
+ WRITE_SIZE, both in KiB, from their separate passes — stamped with the digest of the kernel sources it was measured on
(bench.py quotes the file only while that digest matches).
usage: python tools/update_traffic.py <summary.txt> <name of the summary under profiles/> [rows per launch]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KERNEL = "agg_hash_shape_fixed_kernel<qsx::ShapeTpchQ1, 4, 16, 4, 1, false>"   # (the last argument: no per-wave register groups)
PROBE_KERNEL = "dense_probe_kernel<int, 0, false>"


def counters_of(summary, kernel):
    found, inside = {}, False
    for line in open(summary):
        if not line.startswith(" "):
            inside = kernel in line
            continue
        m = re.match(r"\s+(\w+)\s+n=\s*\d+\s+mean=([0-9.e+]+)", line)
        if inside and m and m.group(1) not in found:
            found[m.group(1)] = float(m.group(2))
    return found


def main():
    summary, committed_as = sys.argv[1], sys.argv[2]
    rows = int(sys.argv[3]) if len(sys.argv) > 3 else 600_000_000
    agg = counters_of(summary, KERNEL)
    fetch, write = agg.get("FETCH_SIZE"), agg.get("WRITE_SIZE")
    if fetch is None or write is None:
        raise SystemExit(f"{summary}: no FETCH_SIZE / WRITE_SIZE for {KERNEL}")
    import bench
    out = {"rows_per_launch": rows, "hbm_bytes_per_launch": int(round((2 * fetch + write) * 1024)), "fetch_size_kib_raw": fetch,
           "write_size_kib_raw": write, "kernel_source_digest": bench.kernel_source_digest(),
           "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/prof_pmc.sh) on {KERNEL}, profiles/{committed_as}; "
                     "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of a wide coalesced stream)"}
    probe = counters_of(summary, PROBE_KERNEL)
    if "FETCH_SIZE" in probe and "WRITE_SIZE" in probe:
        # the probe's key stream is wide and coalesced (doubled like the aggregation's), its head-word reads are narrow and
        # random: the guide calls those uncalibrated, so the figure is quoted with both readings of FETCH_SIZE
        out["probe"] = {"kernel": PROBE_KERNEL, "fetch_size_kib_raw": probe["FETCH_SIZE"], "write_size_kib_raw": probe["WRITE_SIZE"],
                        "hbm_bytes_per_launch": int(round((2 * probe["FETCH_SIZE"] + probe["WRITE_SIZE"]) * 1024)),
                        "hbm_bytes_per_launch_fetch_undoubled": int(round((probe["FETCH_SIZE"] + probe["WRITE_SIZE"]) * 1024)),
                        "tcc_hit": probe.get("TCC_HIT"), "tcc_miss": probe.get("TCC_MISS"), "tcc_req": probe.get("TCC_REQ")}
    with open(os.path.join(ROOT, "profiles", "traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
