#!/usr/bin/env python3
"""Condenses a rocprofv3 --kernel-trace --stats CSV pair into a short per-kernel table (kernel names shortened).

usage: rocprof_summary.py <dir with *_kernel_stats.csv> <out.csv> [note]
"""
import csv
import glob
import re
import sys


def short(name):
    m = re.search(r"(k_[a-z_]+(?:<[^>]*>)?)", name)
    if m:
        return m.group(1)
    name = re.sub(r"\(.*", "", name)
    return name.split("::")[-1][:60]


def main():
    src, out = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    stats = glob.glob(f"{src}/*kernel_stats.csv")[0]
    rows = list(csv.DictReader(open(stats)))
    with open(out, "w", newline="") as f:
        if note:
            f.write(f"# {note}\n")
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "percent", "min_us", "max_us"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], f"{int(r['TotalDurationNs']) / 1e6:.3f}",
                        f"{float(r['AverageNs']) / 1e3:.2f}", r["Percentage"], f"{int(r['MinNs']) / 1e3:.2f}",
                        f"{int(r['MaxNs']) / 1e3:.2f}"])
    print(open(out).read())


if __name__ == "__main__":
    main()
