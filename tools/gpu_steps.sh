#!/bin/bash
# Runs the named steps one after the other on the GPU box, each under its own timeout, logging to gpurun_out/<tag>/.
# A step that times out or is killed ends the sequence (no further GPU work after a hang).
#   tools/gpu_steps.sh <tag> "<name>::<timeout s>::<command>" ...
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  name=${spec%%::*}; rest=${spec#*::}; tmo=${rest%%::*}; cmd=${rest#*::}
  echo "== $name (timeout $tmo s): $cmd"
  start=$(date +%s)
  timeout -k 10 $tmo bash -c "$cmd" > $out/$name.log 2> $out/$name.err
  rc=$?
  echo "== $name rc=$rc $(( $(date +%s) - start )) s"; tail -n 4 $out/$name.log | cut -c1-600
  if [ $rc -ne 0 ]; then tail -n 12 $out/$name.err | cut -c1-400; fi
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "== stopping: $name timed out"; exit 1; fi
done
exit 0
