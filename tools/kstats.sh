#!/bin/bash
# Kernel statistics of the serialised bench (top kernels):  tools/kstats.sh <tag> [VAR=value ...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$tag || exit 1
for v in "$@"; do export "$v"; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/trace -- python3 bench.py --serial --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/$tag/log.txt 2>&1
f=$(ls gpurun_out/$tag/trace/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0].replace("void ", "")
    if n.startswith(("k_pose", "k_prune", "k_topk", "k_corr", "k_describe", "k_orient", "k_pair", "k_zero", "k_copy")):
        print("%-28s calls %4s avg %7.1f us min %7.1f max %7.1f" % (n[:28], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
