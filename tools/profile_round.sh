#!/bin/bash
# One gpurun call: kernel trace + stats, the PMC passes (each its own run, as gpurun requires), their summary, and the un-profiled
# bench line of the same build.    tools/profile_round.sh <tag> [git head]      -> gpurun_out/<tag>/
# rocprofv3 gets `python3 bench.py ...` directly after `--` (no env / bash -c hop: the profiler's preload initialises the GPU).
tag=${1:-prof}; export MAD_GIT_HEAD=${2:-unknown}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$tag || exit 1
B="python3 bench.py --serial --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/trace -- $B --steps 5 --warmup 2 > gpurun_out/$tag/trace.log 2>&1 && \
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/$tag/pmc_fetch -- $B --steps 3 --warmup 1 > gpurun_out/$tag/pmc_fetch.log 2>&1 && \
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/$tag/pmc_write -- $B --steps 3 --warmup 1 > gpurun_out/$tag/pmc_write.log 2>&1 && \
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/$tag/pmc_tcc -- $B --steps 3 --warmup 1 > gpurun_out/$tag/pmc_tcc.log 2>&1 && \
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/$tag/pmc_sq -- $B --steps 3 --warmup 1 > gpurun_out/$tag/pmc_sq.log 2>&1 && \
python tools/pmc_summary.py gpurun_out/$tag/summary.json gpurun_out/$tag/trace gpurun_out/$tag/pmc_fetch gpurun_out/$tag/pmc_write gpurun_out/$tag/pmc_tcc gpurun_out/$tag/pmc_sq > gpurun_out/$tag/summary.txt 2>&1 && \
python bench.py > gpurun_out/$tag/line.json 2> gpurun_out/$tag/line.err
rc=$?
tail -3 gpurun_out/$tag/summary.txt; cut -c1-300 gpurun_out/$tag/line.json
exit $rc
