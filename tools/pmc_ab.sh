#!/bin/bash
# A/B of two builds under the same counters (each library: one kernel trace + two PMC passes), k_describe / k_orient rows only:
#   tools/pmc_ab.sh <tag> <libA.so> <libB.so>      -> gpurun_out/<tag>/{a,b}.txt
tag=$1; A=$2; Bl=$3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$tag || exit 1
B="python3 bench.py --serial --no-cpu-baseline --steps 3 --warmup 1"
for v in a b; do
  if [ $v = a ]; then export MAD_LIB_PATH=$A; else export MAD_LIB_PATH=$Bl; fi
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$tag/${v}_trace -- $B > gpurun_out/$tag/${v}_trace.log 2>&1 && \
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/$tag/${v}_p1 -- $B > gpurun_out/$tag/${v}_p1.log 2>&1 && \
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d gpurun_out/$tag/${v}_p2 -- $B > gpurun_out/$tag/${v}_p2.log 2>&1 && \
  python tools/pmc_summary.py gpurun_out/$tag/$v.json gpurun_out/$tag/${v}_trace gpurun_out/$tag/${v}_p1 gpurun_out/$tag/${v}_p2 > gpurun_out/$tag/$v.txt 2>&1 || exit 1
done
