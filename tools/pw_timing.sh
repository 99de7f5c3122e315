#!/bin/bash
# phase timers (TSS_TIMING=1 build) of the lean 1x1 kernels at the shapes given as "mode K N P" lines on stdin
set -e
cd $GRAFT_REPO_ROOT
TSS_TIMING=1 python -m torch_semantic_segmentation_amd.build > /dev/null
while read mode K N P; do
  echo "== $mode K=$K N=$N P=$P"
  TSS_TIMING=1 python tools/micro_one.py $mode $K $N $P 2>/dev/null
done
