#!/bin/bash
# Kernel statistics of the rehearsed rank of eight on C4 (one GPU doing rank 0's per-step work), lanes serialised:  tools/kstats_rank8.sh <tag>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$tag || exit 1
for v in "$@"; do export "$v"; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/trace -- python3 bench.py --emulate-rank-of 8 --workload c4 --serial --no-cpu-baseline --steps 20 --warmup 2 > gpurun_out/$tag/log.txt 2>&1
f=$(ls gpurun_out/$tag/trace/*/*kernel_stats.csv | head -1)
cp $f gpurun_out/$tag/kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0].replace("void ", "")
    if n.startswith(("k_pose", "k_prune", "k_topk", "k_corr", "k_describe", "k_orient", "k_pair", "k_zero", "k_copy", "k_set", "k_row", "k_shard", "k_anchor", "k_fill")):
        print("%-28s calls %4s avg %7.1f us min %7.1f max %7.1f" % (n[:28], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
tail -1 gpurun_out/$tag/log.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'latency', d['config'].get('latency_ms_single_step'))"
