#!/usr/bin/env python3
"""Trim a rocprofv3 --kernel-trace --stats kernel_stats.csv to a readable per-kernel table (names shortened)."""
import csv
import re
import sys


def short(n):
    n = re.sub(r"\(.*", "", n)
    if "bella::" not in n:
        n = re.sub(r"<.*", "<...>", n)                       # library kernels: template arguments are noise
    n = n.replace("void ", "").replace("bella::", "")
    return n[-90:]


def main(path, top=25):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    print("%-62s %7s %14s %14s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for r in rows[:top]:
        print("%-62s %7s %14.1f %14.2f %8s" % (short(r["Name"])[:62], r["Calls"], float(r["TotalDurationNs"]) / 1e3,
                                             float(r["AverageNs"]) / 1e3, r["Percentage"]))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25)
