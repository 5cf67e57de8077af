#!/bin/bash
# Counters of the match kernels on C4 (one GPU, lanes serialised): tools/pmc_c4.sh  -> gpurun_out/pmc_c4/s.{json,txt}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc_c4
B="python3 bench.py --serial --no-cpu-baseline --steps 2 --warmup 1 --repeat-anchors --workload c4"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc_c4/trace -- $B > gpurun_out/pmc_c4/trace.log 2>&1 && \
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_c4/p2 -- $B > gpurun_out/pmc_c4/p2.log 2>&1 && \
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d gpurun_out/pmc_c4/p3 -- $B > gpurun_out/pmc_c4/p3.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_c4/s.json gpurun_out/pmc_c4/trace gpurun_out/pmc_c4/p2 gpurun_out/pmc_c4/p3 > gpurun_out/pmc_c4/s.txt 2>&1
grep "k_pose_bounds\|k_pose_lds\|k_pose_setup\|k_corr\|k_pair" gpurun_out/pmc_c4/s.txt | cut -c1-700
rm -rf gpurun_out/pmc_c4/p2 gpurun_out/pmc_c4/p3
