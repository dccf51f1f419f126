#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc counter_collection CSVs (one directory per pass) into per-kernel totals.

usage: pmc_summary.py <out.csv> <dir> [<dir> ...]
Counters are summed over all dispatches of a kernel; FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them.
"""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(k_[a-z_]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else re.sub(r"\(.*", "", name).split("::")[-1][:50]


def main():
    out = sys.argv[1]
    tot = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(int)
    for d in sys.argv[2:]:
        for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
            seen = set()
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
                key = (k, r["Dispatch_Id"])
                if key not in seen:
                    seen.add(key)
            for k, _ in seen:
                calls[k] = max(calls[k], sum(1 for kk, _ in seen if kk == k))
    counters = sorted({c for k in tot for c in tot[k]})
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "dispatches"] + counters)
        for k in sorted(tot):
            if not k.startswith("k_"):
                continue
            w.writerow([k, calls[k]] + [f"{tot[k].get(c, 0.0):.0f}" for c in counters])
    print(open(out).read())


if __name__ == "__main__":
    main()
