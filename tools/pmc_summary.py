#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel, mean counter value per dispatch.
usage: pmc_summary.py <dir with *_counter_collection.csv> [kernel-substring ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    want = sys.argv[2:]
    files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name") or row.get("Kernel Name") or ""
                if want and not any(w in name for w in want):
                    continue
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for name, counters in acc.items():
        print(name[:100])
        for c, vals in sorted(counters.items()):
            print(f"   {c:32s} n={len(vals):3d} mean={sum(vals) / len(vals):.6g}")


if __name__ == "__main__":
    main()
