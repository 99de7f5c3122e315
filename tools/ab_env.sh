#!/bin/bash
# A/B of one environment tunable: tools/ab_env.sh VAR v1 v2 ...   (FastSCNN train step, ms/step per value)
cd $GRAFT_REPO_ROOT
var=$1; shift
for v in "$@"; do
  env $var=$v python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-roofline --no-extras > gpurun_out/ab.json 2>/dev/null
  echo "$var=$v $(grep -o 'median_ms_per_step.: [0-9.]*' gpurun_out/ab.json)"
done
