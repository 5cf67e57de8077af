#!/bin/bash
# k_describe alone (the default) against k_describe + k_describe_ball (MAD_BALL=1) under the same counters, serialised bench:
#   tools/pmc_ball_ab.sh <tag>      -> gpurun_out/<tag>/{rowwise,ball}.{json,txt}
# Each counter group is its own run (gpurun refuses --pmc together with the trace domains other than the kernel trace).
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$tag || exit 1
B="python3 bench.py --serial --no-cpu-baseline --steps 3 --warmup 1 --repeat-anchors"
for v in rowwise ball; do
  if [ $v = ball ]; then export MAD_BALL=1; else export MAD_BALL=0; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/${v}_trace -- $B > gpurun_out/$tag/${v}_trace.log 2>&1 && \
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/$tag/${v}_p0 -- $B > gpurun_out/$tag/${v}_p0.log 2>&1 && \
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/$tag/${v}_p0w -- $B > gpurun_out/$tag/${v}_p0w.log 2>&1 && \
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/$tag/${v}_p1 -- $B > gpurun_out/$tag/${v}_p1.log 2>&1 && \
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/$tag/${v}_p2 -- $B > gpurun_out/$tag/${v}_p2.log 2>&1 && \
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d gpurun_out/$tag/${v}_p3 -- $B > gpurun_out/$tag/${v}_p3.log 2>&1 && \
  python tools/pmc_summary.py gpurun_out/$tag/$v.json gpurun_out/$tag/${v}_trace gpurun_out/$tag/${v}_p0 gpurun_out/$tag/${v}_p0w gpurun_out/$tag/${v}_p1 gpurun_out/$tag/${v}_p2 gpurun_out/$tag/${v}_p3 > gpurun_out/$tag/$v.txt 2>&1 || { tail -3 gpurun_out/$tag/${v}_*.log; exit 1; }
  grep "k_describe" gpurun_out/$tag/$v.txt | cut -c1-400
  rm -rf gpurun_out/$tag/${v}_p0 gpurun_out/$tag/${v}_p0w gpurun_out/$tag/${v}_p1 gpurun_out/$tag/${v}_p2 gpurun_out/$tag/${v}_p3
done
