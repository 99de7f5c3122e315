#!/bin/bash
# A/B: weight gradient of a layer on a side stream next to its input gradient, for layers in a size window (elements of x + y)
cd $GRAFT_REPO_ROOT
run() { env "$@" python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-roofline --no-extras > gpurun_out/ab.json 2>gpurun_out/ab.err; echo "$* $(grep -o 'median_ms_per_step.: [0-9.]*' gpurun_out/ab.json)"; }
run TSS_OVERLAP_WGRAD=0
run TSS_OVERLAP_WGRAD=1 TSS_OVERLAP_MIN=30000000 TSS_OVERLAP_MAX=9000000000
run TSS_OVERLAP_WGRAD=1 TSS_OVERLAP_MIN=60000000 TSS_OVERLAP_MAX=9000000000
run TSS_OVERLAP_WGRAD=1 TSS_OVERLAP_MIN=100000000 TSS_OVERLAP_MAX=9000000000
run TSS_OVERLAP_WGRAD=1 TSS_OVERLAP_MIN=0 TSS_OVERLAP_MAX=9000000000
