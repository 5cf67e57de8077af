#!/bin/bash
# A/B of two builds, memory-side counters of k_describe:  tools/pmc_ab2.sh <tag> <libA.so> <libB.so>
# (few counters per pass: a set the hardware cannot collect at once aborts rocprofv3, and its shutdown then hangs)
tag=$1; A=$2; Bl=$3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$tag || exit 1
B="python3 bench.py --serial --no-cpu-baseline --steps 3 --warmup 1"
T="timeout -k 10 120"
for v in a b; do
  if [ $v = a ]; then export MAD_LIB_PATH=$A; else export MAD_LIB_PATH=$Bl; fi
  $T rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$tag/${v}_trace -- $B > gpurun_out/$tag/${v}_trace.log 2>&1 && \
  $T rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/$tag/${v}_p0 -- $B > gpurun_out/$tag/${v}_p0.log 2>&1 && \
  $T rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/$tag/${v}_p1 -- $B > gpurun_out/$tag/${v}_p1.log 2>&1 && \
  $T rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM --kernel-trace --output-format csv -d gpurun_out/$tag/${v}_p2 -- $B > gpurun_out/$tag/${v}_p2.log 2>&1 && \
  python tools/pmc_summary.py gpurun_out/$tag/$v.json gpurun_out/$tag/${v}_trace gpurun_out/$tag/${v}_p0 gpurun_out/$tag/${v}_p1 gpurun_out/$tag/${v}_p2 > gpurun_out/$tag/$v.txt 2>&1 || { echo "pass failed ($v)"; exit 1; }
  echo "done $v"
done
