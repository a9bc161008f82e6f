#!/usr/bin/env python3
"""Every dispatch of the kernels whose name contains argv[2], in launch order: start (ms since the first), duration (us).
usage: rocpd_kernel_list.py <rocpd .db> <substring> [more substrings]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
scol = [r[1] for r in db.execute(f"pragma table_info({sym})")]
name_col = "kernel_name" if "kernel_name" in scol else ("display_name" if "display_name" in scol else scol[-1])
rows = list(db.execute(f"select s.{name_col}, d.start, d.end from {disp} d join {sym} s on d.kernel_id = s.id order by d.start"))
t0 = rows[0][1] if rows else 0
for name, start, end in rows:
    if any(k in name for k in sys.argv[2:]):
        print(f"{(start - t0) / 1e6:10.3f} ms  {(end - start) / 1e3:9.1f} us  {name[:70]}")
