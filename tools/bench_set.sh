#!/bin/bash
# The bench lines a round compares: C3 (default), the rehearsed rank of eight on C4, C4 and C5 on one GPU.  tools/bench_set.sh <tag>
tag=${1:-bs}; mkdir -p gpurun_out/$tag
python bench.py --no-cpu-baseline > gpurun_out/$tag/c3.json 2> gpurun_out/$tag/c3.err || { tail -5 gpurun_out/$tag/c3.err; exit 1; }
python bench.py --no-cpu-baseline --workload c4 --emulate-rank-of 8 > gpurun_out/$tag/rank8.json 2> gpurun_out/$tag/rank8.err || { tail -5 gpurun_out/$tag/rank8.err; exit 1; }
python bench.py --no-cpu-baseline --workload c4 --steps 10 > gpurun_out/$tag/c4.json 2> gpurun_out/$tag/c4.err || { tail -5 gpurun_out/$tag/c4.err; exit 1; }
python bench.py --no-cpu-baseline --workload c3clean > gpurun_out/$tag/c3clean.json 2> gpurun_out/$tag/c3clean.err || { tail -5 gpurun_out/$tag/c3clean.err; exit 1; }
python bench.py --no-cpu-baseline --workload c5 --emulate-rank-of 8 --steps 6 --warmup 2 > gpurun_out/$tag/c5rank8.json 2> gpurun_out/$tag/c5rank8.err || { tail -5 gpurun_out/$tag/c5rank8.err; exit 1; }
python bench.py --no-cpu-baseline --workload c5 --steps 6 --warmup 2 > gpurun_out/$tag/c5.json 2> gpurun_out/$tag/c5.err || { tail -5 gpurun_out/$tag/c5.err; exit 1; }
python - $tag <<'PY'
import json, sys
t = sys.argv[1]
for n in ("c3", "c3clean", "rank8", "c4", "c5rank8", "c5"):
    d = json.load(open("gpurun_out/%s/%s.json" % (t, n)))
    print("%-6s %.4g corr/s  %.4f ms/step  latency %s  host %s" % (n, d["value"], d["ms_per_step"], d.get("latency_ms_single_step"), d["roofline"].get("host_ms_per_step_timed_region")))
PY
