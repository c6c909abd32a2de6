#!/bin/bash
# Run several GPU steps in ONE gpurun call: tools/gpu_steps.sh <tag> "<seconds> <name> <command...>" ...
# Each step's output goes to gpurun_out/<tag>/<name>.txt.  A failing step does not stop the next one; a step that is
# killed at its limit (124 / 137) DOES -- nothing else is started on a GPU that may be wedged.
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for step in "$@"; do
  set -- $step
  secs=$1; name=$2; shift 2
  echo "== $name ($secs s): $*"
  start=$(date +%s)
  timeout -k 10 $secs bash -c "$*" > $O/$name.txt 2>&1
  rc=$?
  echo "   rc=$rc in $(( $(date +%s) - start )) s; tail:"; tail -n 4 $O/$name.txt | cut -c1-300
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name was killed at its limit: stopping"; exit $rc; fi
done
