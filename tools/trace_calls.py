#!/usr/bin/env python3
"""Durations of the calls of one kernel, in call order, from a rocprofv3 kernel trace, folded into groups:
   tools/trace_calls.py <dir> <kernel name prefix> [group size]  -> mean duration per position in the group"""
import csv, glob, sys
d, name = sys.argv[1], sys.argv[2]
grp = int(sys.argv[3]) if len(sys.argv) > 3 else 1
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].replace("void ", "").startswith(name)]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print("%d calls of %s" % (len(dur), name))
for k in range(grp):
    v = dur[k::grp]
    if v:
        print("  position %d: mean %.1f us  min %.1f  max %.1f  (n=%d)" % (k, sum(v) / len(v), min(v), max(v), len(v)))
