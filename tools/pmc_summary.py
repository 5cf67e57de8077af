#!/usr/bin/env python3
"""Summarise rocprofv3 output directories into small JSON/CSV files for profiles/.

  python tools/pmc_summary.py <out.json> <kernel_trace_dir> [<pmc_dir> ...]

Per kernel of this library: launches, average duration (kernel trace) and the per-launch average of every
PMC counter found (FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them; `*_bytes` applies the
unit, FETCH additionally the x2 gfx950 correction for wide streaming reads is NOT applied because the
kernels here read 16-byte texels and 64-byte sectors, not 1-KiB wave rows -- see MI355X_MICROARCH.md HBM)."""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    n = name.split("(")[0].replace("void ", "").strip()
    return n


def main():
    out, trace_dir, pmc_dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    res = collections.defaultdict(dict)
    for f in glob.glob(os.path.join(trace_dir, "**", "*kernel_trace.csv"), recursive=True):
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            n = short(r["Kernel_Name"])
            if n.startswith("k_"):
                dur[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for n, v in dur.items():
            res[n]["launches"] = len(v)
            res[n]["avg_us"] = sum(v) / len(v) / 1e3
            res[n]["total_us"] = sum(v) / 1e3
    for d in pmc_dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(f)):
                n = short(r["Kernel_Name"])
                if n.startswith("k_"):
                    acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for n, cs in acc.items():
                for c, v in cs.items():
                    res[n][c + "_avg"] = sum(v) / len(v)
                    if c in ("FETCH_SIZE", "WRITE_SIZE"):
                        res[n][c.lower() + "_bytes_avg"] = 1024.0 * sum(v) / len(v)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench      # source_fingerprint(): which build of the kernels these counters belong to (bench.py checks it before quoting them)
    doc = {"_meta": {"source_fingerprint": bench.source_fingerprint(), "git": os.environ.get("MAD_GIT_HEAD"),
                     "command": "rocprofv3 --kernel-trace --stats / --pmc ... -- python3 bench.py --serial (tools/profile_round.sh)"}}
    doc.update({k: res[k] for k in sorted(res, key=lambda k: -res[k].get("total_us", 0))})
    with open(out, "w") as fh:
        json.dump(doc, fh, indent=1)
    for k in sorted(res, key=lambda k: -res[k].get("total_us", 0))[:12]:
        print(k, {a: round(b, 1) for a, b in res[k].items()})


if __name__ == "__main__":
    main()
