#!/bin/bash
# Run a list of GPU steps one after another on a gpurun box; every step under its own timeout, logs under gpurun_out/.
# A step that is killed (timeout / signal) ends the call: no further GPU work is started after a hang.
#   tools/gpu_steps.sh "name1|seconds|command ..." "name2|seconds|command ..."
mkdir -p gpurun_out
for spec in "$@"; do
  name="${spec%%|*}"; rest="${spec#*|}"; secs="${rest%%|*}"; cmd="${rest#*|}"
  echo "=== $name (limit ${secs}s): $cmd"
  start=$(date +%s)
  timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2> "gpurun_out/$name.err"
  rc=$?
  echo "=== $name rc=$rc in $(( $(date +%s) - start ))s"
  tail -n 6 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -ge 128 ]; then
    echo "=== $name was killed; stopping here"; tail -n 20 "gpurun_out/$name.err"; exit $rc
  fi
done
exit 0
