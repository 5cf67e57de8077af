#!/usr/bin/env python3
"""What the device did during the overlapped steps of a bench run, from a rocprofv3 kernel trace (csv):
how long at least one kernel was running, how long two or more were, the idle gaps, and per kernel the time it ran
ALONE on the device against the time it shared it.    tools/timeline.py <kernel_trace.csv> [first_fraction last_fraction]"""
import csv
import gzip
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    for p in ("void ", "(anonymous namespace)::"):
        name = name.replace(p, "")
    return name.strip()


def main():
    path = sys.argv[1]
    lo_f = float(sys.argv[2]) if len(sys.argv) > 2 else 0.35
    hi_f = float(sys.argv[3]) if len(sys.argv) > 3 else 0.95
    op = gzip.open if path.endswith(".gz") else open
    rows = []
    with op(path, "rt") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")))
    hot = sorted(r for r in rows if r[2].startswith("k_describe"))
    if not hot:
        print("no k_describe launches in the trace")
        return
    # the window: between two describe launches well inside the timed steps
    t0 = hot[int(len(hot) * lo_f)][0]
    t1 = hot[int(len(hot) * hi_f)][0]
    n_desc = sum(1 for r in hot if t0 <= r[0] < t1)
    ev = []
    for s, e, n, q in rows:
        s2, e2 = max(s, t0), min(e, t1)
        if e2 > s2:
            ev.append((s2, 1, n))
            ev.append((e2, -1, n))
    ev.sort()
    span = t1 - t0
    depth_time = defaultdict(int)
    alone, shared = defaultdict(int), defaultdict(int)
    active = defaultdict(int)
    prev, depth = t0, 0
    for t, d, n in ev:
        dt = t - prev
        if dt > 0:
            depth_time[min(depth, 6)] += dt
            names = [k for k, v in active.items() if v > 0]
            for k in names:
                (alone if depth == 1 else shared)[k] += dt
        prev = t
        depth += d
        active[n] += d
    depth_time[min(depth, 6)] += t1 - prev
    queues = defaultdict(int)
    for s, e, n, q in rows:
        s2, e2 = max(s, t0), min(e, t1)
        if e2 > s2:
            queues[q] += e2 - s2
    print("window %.3f ms, %d k_describe launches in it" % (span / 1e6, n_desc))
    print("kernels running at once -> share of the window")
    for d in sorted(depth_time):
        print("  %s%d: %5.1f %%" % (">=" if d == 6 else "  ", d, 100.0 * depth_time[d] / span))
    print("busy time per queue (ms):", {q: round(v / 1e6, 3) for q, v in sorted(queues.items())})
    print("%-28s %10s %10s %10s" % ("kernel", "alone ms", "shared ms", "sum of durations ms"))
    tot = defaultdict(int)
    for s, e, n, q in rows:
        s2, e2 = max(s, t0), min(e, t1)
        if e2 > s2:
            tot[n] += e2 - s2
    for n in sorted(tot, key=lambda k: -tot[k])[:16]:
        print("%-28s %10.3f %10.3f %10.3f" % (n[:28], alone[n] / 1e6, shared[n] / 1e6, tot[n] / 1e6))
    print("sum of all kernel durations / window = %.2f (average number of kernels in flight)" % (sum(tot.values()) / span))


if __name__ == "__main__":
    main()
