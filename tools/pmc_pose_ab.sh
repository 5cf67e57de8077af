#!/bin/bash
# k_pose_bounds (vector form, MAD_POSE_MX=0) against k_pose_bounds_mx (the default) under the same counters, serialised bench:
#   tools/pmc_pose_ab.sh <tag>      -> gpurun_out/<tag>/{vec,mx}.{json,txt}
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$tag || exit 1
B="python3 bench.py --serial --no-cpu-baseline --steps 3 --warmup 1 --repeat-anchors"
for v in vec mx; do
  if [ $v = mx ]; then export MAD_POSE_MX=1; else export MAD_POSE_MX=0; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/${v}_trace -- $B > gpurun_out/$tag/${v}_trace.log 2>&1 && \
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/$tag/${v}_p2 -- $B > gpurun_out/$tag/${v}_p2.log 2>&1 && \
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d gpurun_out/$tag/${v}_p3 -- $B > gpurun_out/$tag/${v}_p3.log 2>&1 && \
  rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --kernel-trace --output-format csv -d gpurun_out/$tag/${v}_p4 -- $B > gpurun_out/$tag/${v}_p4.log 2>&1
  python tools/pmc_summary.py gpurun_out/$tag/$v.json gpurun_out/$tag/${v}_trace gpurun_out/$tag/${v}_p2 gpurun_out/$tag/${v}_p3 gpurun_out/$tag/${v}_p4 > gpurun_out/$tag/$v.txt 2>&1 || { tail -3 gpurun_out/$tag/${v}_*.log; }
  grep "k_pose_bounds" gpurun_out/$tag/$v.txt | cut -c1-600
  rm -rf gpurun_out/$tag/${v}_p2 gpurun_out/$tag/${v}_p3 gpurun_out/$tag/${v}_p4
done
