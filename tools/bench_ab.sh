#!/bin/bash
# Quick A/B of library builds on the default workload:  tools/bench_ab.sh libA.so libB.so ...   (paths under mad_amd/csrc/ or absolute)
for v in "$@"; do
  case $v in /*) p=$v;; *) p=mad_amd/csrc/$v;; esac
  MAD_LIB_PATH=$p python bench.py --no-cpu-baseline --steps 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v', 'ms/step %.4f' % d['ms_per_step'], 'serial', r['timing'].split('(')[1].split(')')[0], {k: round(v, 4) for k, v in r['kernel_ms_per_step'].items()}, d['config'].get('topk_agrees_with_cpu_oracle'))"
done
