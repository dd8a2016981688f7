#!/usr/bin/env python3
"""dev: the last <ms> milliseconds of a rocprofv3 kernel + memory-copy trace as one time-ordered list
usage: trace_dump.py <dir> [ms, default 40] [min_us, default 20]"""
import csv, glob, os, sys
d = sys.argv[1]; span = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0; min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0].replace("void ", "")[:50], r.get("Stream_Id", r.get("Queue_Id", ""))))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r["Direction"].replace("MEMORY_COPY_", ""), r.get("Stream_Id", "")))
ev.sort()
t_end = max(e[1] for e in ev)
t0 = t_end - int(span * 1e6)
for s, e, n, q in ev:
    if e < t0 or (e - s) / 1e3 < min_us:
        continue
    print("%10.1f %10.1f %9.1f  s%-4s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n))
