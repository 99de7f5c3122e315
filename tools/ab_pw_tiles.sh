#!/bin/bash
# A/B of the tile-size thresholds of the single-chunk 1x1 kernels: tools/ab_pw_tiles.sh "bwd fwd" ...
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  set -- $cfg
  TSS_PW_BWD_SMALL=$1 TSS_PW_FWD_SMALL=$2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/ab.json 2>/dev/null
  echo "bwd_thr=$1 fwd_thr=$2 $(grep -o 'ms_per_step.: [0-9.]*' gpurun_out/ab.json)"
done
