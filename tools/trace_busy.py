#!/usr/bin/env python3
"""Busy fraction of the device in a rocprofv3 kernel trace: union of the kernel intervals of this library over the span
of the last `n` bench steps (a step starts at a k_copy_words/k_orient burst; here simply the last `frac` of the trace)."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
# the timed region: between the 3rd-from-last ... use k_orient launches on the map (longest k_orient per step) as step markers
starts = [s for s, e, n in rows if n.startswith("k_topk_sort")]
print("k_topk_sort launches", len(starts))
# take the window covering the last 40 top-k sorts (10 steps x 4 matches) if available
k = min(len(starts), 40)
t0 = starts[-k]
t1 = rows[-1][1]
sel = [(s, e) for s, e, n in rows if e > t0 and n.startswith("k_") or n.startswith("void k_")]
sel = [(max(s, t0), e) for s, e in sel if e > t0]
sel.sort()
busy, cur_s, cur_e = 0, None, None
for s, e in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("window %.3f ms, device busy (union of kernels) %.3f ms = %.1f %%" % ((t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0)))
gaps = []
cur_e = None
for s, e in sel:
    if cur_e is not None and s > cur_e: gaps.append(s - cur_e)
    cur_e = e if cur_e is None else max(cur_e, e)
gaps.sort(reverse=True)
print("largest gaps (us):", [round(g / 1e3, 1) for g in gaps[:20]])
print("gap total %.3f ms in %d gaps" % (sum(gaps) / 1e6, len(gaps)))
