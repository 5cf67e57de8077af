#!/usr/bin/env python3
"""Per-kernel table of a rocprofv3 --kernel-trace --stats run: tools/kstats_all.py <dir with *_kernel_stats.csv> [steps]"""
import csv, glob, sys
d = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
f = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows:
    n = r["Name"].split("(")[0].replace("void ", "")[:44]
    line = "%-46s calls %5s avg %8.1f us  share %5.2f%%" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot)
    if steps:
        line += "  per step %7.1f us (%4.1f launches)" % (float(r["TotalDurationNs"]) / 1e3 / steps, float(r["Calls"]) / steps)
    print(line)
