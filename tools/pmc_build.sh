#!/bin/bash
# SQ instruction counters of the build stage alone (tools/probe_build.py), default against an environment variant:
#   tools/pmc_build.sh <tag> [VAR=value]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$tag || exit 1
for v in "$@"; do export "$v"; done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/$tag/pmc -- python3 tools/probe_build.py c3 3 > gpurun_out/$tag/log.txt 2>&1
python3 - <<'PY' gpurun_out/$tag
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/pmc/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "describe" in k or "k_orient<" in k:
        print(k, {c: round(sum(x) / len(x) / 1e6, 2) for c, x in v.items()}, "launches", len(next(iter(v.values()))))
PY
