#!/usr/bin/env python3
"""Device busy fraction inside the timed region of a rocprofv3 kernel trace of `bench.py --steps K --warmup W`:
union of the kernel intervals, largest gaps, mean number of kernels in flight.  usage: trace_busy.py <dir> K W"""
import csv, glob, sys
K, W = int(sys.argv[2]), int(sys.argv[3])
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
sorts = [(s, e) for s, e, n in rows if n.startswith("k_topk_sort")]
prime = 4      # bench.py primes both set groups with 2 steps each
first = 4 * (prime + W)
t0, t1 = sorts[first - 1][1], sorts[first + 4 * K - 1][1]
sel = sorted((max(s, t0), min(e, t1)) for s, e, n in rows if e > t0 and s < t1)
busy, cs, ce, gaps = 0, None, None, []
for s, e in sel:
    if ce is None or s > ce:
        if ce is not None:
            busy += ce - cs
            gaps.append(s - ce)
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print("window %.3f ms (%.3f ms/step), busy %.3f ms = %.1f%%" % ((t1 - t0) / 1e6, (t1 - t0) / 1e6 / K, busy / 1e6, 100 * busy / (t1 - t0)))
gaps.sort(reverse=True)
print("largest gaps us:", [round(g / 1e3, 1) for g in gaps[:16]])
print("gaps > 20 us: %d totalling %.3f ms" % (sum(1 for g in gaps if g > 20000), sum(g for g in gaps if g > 20000) / 1e6))
ev = []
for s, e in sel:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
cur, last, area = 0, None, 0
for t, d in ev:
    if last is not None: area += cur * (t - last)
    cur += d; last = t
print("mean kernels in flight while busy: %.2f" % (area / busy))
