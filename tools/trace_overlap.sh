#!/bin/bash
# Kernel trace of the OVERLAPPED bench (lanes on their own streams, 3 steps in flight) for tools/timeline.py.
#   tools/trace_overlap.sh <tag> [extra bench args]   -> gpurun_out/<tag>/trace/.../*_kernel_trace.csv
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$tag || exit 1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$tag/trace -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" > gpurun_out/$tag/trace.log 2>&1
rc=$?
f=$(ls gpurun_out/$tag/trace/*/*_kernel_trace.csv | head -1)
head -2 $f | cut -c1-600
python tools/timeline.py $f > gpurun_out/$tag/timeline.txt 2>&1; tail -40 gpurun_out/$tag/timeline.txt
# keep the merge-back small: the trace itself stays on the box unless it is small
gzip -9 $f
exit $rc
