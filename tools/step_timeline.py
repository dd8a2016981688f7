#!/usr/bin/env python3
"""One step of the hot path as the GPU saw it: every kernel of the LAST step in a rocprofv3 --kernel-trace CSV
(a step = from one k_unit_key launch to the next), its duration and the idle gap in front of it.
usage: step_timeline.py <rocprof output dir> [marker-kernel, default k_unit_key]"""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "k_unit_key"
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not f:
        print("no kernel_trace.csv under", d)
        return
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(marker) or (" " + marker) in r["Kernel_Name"]]
    if len(starts) < 2:
        print("fewer than two steps in the trace")
        return
    a, b = starts[-2], starts[-1]
    t_prev = int(rows[a]["Start_Timestamp"])
    t0 = t_prev
    tot_k = tot_gap = 0
    print("%-86s %10s %10s" % ("kernel (last complete step)", "gap_us", "dur_us"))
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        name = name if len(name) <= 86 else name[:83] + "..."
        gap = max(0, s - t_prev)
        print("%-86s %10.1f %10.1f" % (name, gap / 1e3, (e - s) / 1e3))
        tot_k += e - s
        tot_gap += gap
        t_prev = max(t_prev, e)
    print("step: %.2f ms from first kernel start to next step's first kernel; kernels %.2f ms, gaps %.2f ms (the last gap, to the next step, is the host's restore copy + sync: %.1f us)"
          % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e6, tot_k / 1e6, tot_gap / 1e6, (int(rows[b]["Start_Timestamp"]) - t_prev) / 1e3))


if __name__ == "__main__":
    main()
