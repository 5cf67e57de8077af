#!/bin/bash
# Kernel statistics of the rehearsed rank of an 8-GPU C4 step (serialised lanes):  tools/trace_rank.sh <tag>
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$tag || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/trace -- python3 bench.py --serial --no-cpu-baseline --emulate-rank-of 8 --steps 40 --warmup 4 > gpurun_out/$tag/log.txt 2>&1
rc=$?
f=$(ls gpurun_out/$tag/trace/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.3f ms" % (tot / 1e6))
for r in rows[:60]:
    print("%-46s calls %5s  avg %8.1f us  total %7.3f ms  %5.1f %%" % (r["Name"].split("(")[0][-46:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
exit $rc
