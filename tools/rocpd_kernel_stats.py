#!/usr/bin/env python3
"""Per-kernel duration summary from a rocprofv3 rocpd sqlite database
(rocprofv3 --kernel-trace --stats writes <name>_results.db).  Prints a table
like rocprofv3's kernel_stats.csv: name, calls, total/avg/min/max (us), %."""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in db.execute(f"pragma table_info({disp})")]
    scol = [r[1] for r in db.execute(f"pragma table_info({sym})")]
    name_col = "kernel_name" if "kernel_name" in scol else ("display_name" if "display_name" in scol else scol[-1])
    q = f"select s.{name_col}, d.start, d.end from {disp} d join {sym} s on d.kernel_id = s.id"
    stats = {}
    for name, start, end in db.execute(q):
        st = stats.setdefault(name, [0, 0.0, 1e30, 0.0])
        dur = (end - start) / 1e3
        st[0] += 1
        st[1] += dur
        st[2] = min(st[2], dur)
        st[3] = max(st[3], dur)
    total = sum(s[1] for s in stats.values()) or 1.0
    print(f"{'kernel':90s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for name, (n, tot, mn, mx) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        print(f"{name[:90]:90s} {n:6d} {tot:12.1f} {tot / n:10.1f} {mn:10.1f} {mx:10.1f} {100 * tot / total:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
