#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into the small summaries kept under profiles/.

  prof_summary.py stats <dir>      -> top kernels of *_kernel_stats.csv
  prof_summary.py pmc <dir>        -> per (kernel, counter): dispatches, mean, sum  (*_counter_collection.csv)
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(d, suffix):
    r = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return r[0] if r else None


def short(name):
    name = name.split("(")[0]
    return name if len(name) < 90 else name[:87] + "..."


def main():
    mode, d = sys.argv[1], sys.argv[2]
    if mode == "stats":
        f = find(d, "kernel_stats.csv")
        if not f:
            print("no kernel_stats.csv under", d)
            return
        rows = list(csv.DictReader(open(f)))
        print("%-90s %8s %14s %12s %8s" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
        for r in rows[:25]:
            print("%-90s %8s %14s %12.0f %8s" % (short(r["Name"]), r["Calls"], r["TotalDurationNs"],
                                                 float(r["AverageNs"]), r["Percentage"]))
    else:
        f = find(d, "counter_collection.csv")
        if not f:
            print("no counter_collection.csv under", d)
            return
        acc = defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            kn = short(r["Kernel_Name"]).replace("void ", "")
            if not (kn.startswith("k_") or "rc_" in kn or kn.startswith("gather")):
                continue
            a = acc[(kn, r["Counter_Name"])]
            v = float(r["Counter_Value"])
            a[0] += 1
            a[1] += v
            if len(a) < 3:
                a.append(v)
            else:
                a[2] = max(a[2], v)
        print("%-40s %-28s %10s %20s %20s" % ("kernel", "counter", "dispatches", "largest dispatch", "sum"))
        for (kn, cn), (n, s, mx) in sorted(acc.items()):
            print("%-40s %-28s %10d %20.1f %20.1f" % (kn, cn, n, mx, s))


if __name__ == "__main__":
    main()
