#!/bin/bash
# bench.py without the CPU baseline; prints the headline and the per-kernel ms/step (development helper)
python bench.py --no-cpu-baseline "$@" > gpurun_out/qb.json 2> gpurun_out/qb.err || { tail -20 gpurun_out/qb.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/qb.json"))
print("%.4g corr/s  %.4f ms/step  %s" % (d["value"], d["ms_per_step"], {k: round(v, 4) for k, v in d["roofline"]["kernel_ms_per_step"].items()}))
print(d["roofline"]["timing"])
PY
