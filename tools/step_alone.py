#!/usr/bin/env python3
"""One step submitted alone, from a rocprofv3 kernel trace of a bench run: the launches of the isolated steps bench.py times for
latency_ms_single_step (clusters of kernels between idle gaps), one of them listed launch by launch with its start offset, duration and queue.
    tools/step_alone.py <kernel_trace.csv[.gz]>"""
import csv
import gzip
import sys


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def main():
    path = sys.argv[1]
    op = gzip.open if path.endswith(".gz") else open
    rows = []
    with op(path, "rt") as f:
        for r in csv.DictReader(f):
            n = short(r["Kernel_Name"])
            if n.startswith("k_"):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "?")))
    rows.sort()
    clusters, cur, end = [], [], 0
    for r in rows:
        if cur and r[0] - end > 150_000:      # 150 us of idle device: a new cluster
            clusters.append(cur)
            cur = []
        cur.append(r)
        end = max(end, r[1]) if cur[:-1] else r[1]
    if cur:
        clusters.append(cur)
    steps = [c for c in clusters if 40 <= len(c) <= 80 and sum(1 for r in c if r[2].startswith("k_describe")) == 5]
    print("%d clusters, %d look like one step alone (5 describe launches)" % (len(clusters), len(steps)))
    if not steps:
        return
    spans = sorted((max(r[1] for r in c) - c[0][0]) / 1e3 for c in steps)
    print("device span of a step alone: median %.0f us (min %.0f, max %.0f)" % (spans[len(spans) // 2], spans[0], spans[-1]))
    c = steps[len(steps) // 2]
    t0 = c[0][0]
    busy, last = 0, t0
    for s, e, n, q in c:
        busy += max(0, e - max(s, last))
        last = max(last, e)
    print("device busy (at least one kernel) %.0f us of %.0f" % (busy / 1e3, (last - t0) / 1e3))
    for s, e, n, q in c:
        print("  %8.1f  %7.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n))


if __name__ == "__main__":
    main()
