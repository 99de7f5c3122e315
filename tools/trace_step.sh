#!/bin/bash
# rocprofv3 kernel trace of bench.py's HIP-graph replays -> gpurun_out/gaps.txt (per-kernel totals of the last step)
# and gpurun_out/timeline.txt (start offset, duration, name of every kernel of that step).  Extra args go to bench.py.
set -e
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/trace
rocprofv3 --kernel-trace -d $root/gpurun_out/trace --output-format csv -- python3 $root/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-extras "$@" > /dev/null 2>&1
cd $root
f=$(ls gpurun_out/trace/*/*kernel_trace.csv | head -1)
python3 tools/trace_gaps.py $f gpurun_out/timeline.txt > gpurun_out/gaps.txt
rm -rf gpurun_out/trace
