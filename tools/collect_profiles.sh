#!/bin/bash
# Everything profiles/ holds for one round, in one GPU call: tools/collect_profiles.sh r01
set -e
tag=${1:-r01}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/profiles_$tag
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# 1. per-kernel time of the bench command itself
rocprofv3 --kernel-trace --stats -d $out/stats --output-format csv -- python3 $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/${tag}_bench_under_rocprof.json 2> $out/stats.err
cp $out/stats/*/*kernel_stats.csv $out/${tag}_bench_kernel_stats.csv
cp $out/stats/*/*domain_stats.csv $out/${tag}_bench_domain_stats.csv
rm -rf $out/stats
echo "stats done"
# 2. HBM traffic counters: two counter-only passes over un-captured steps (one dispatch per kernel launch)
rocprofv3 --pmc FETCH_SIZE -d $out/fetch --output-format csv -- python3 $root/bench.py --steps 2 --warmup 1 --graph off --no-cpu-baseline --no-roofline > /dev/null 2> $out/fetch.err
rocprofv3 --pmc WRITE_SIZE -d $out/write --output-format csv -- python3 $root/bench.py --steps 2 --warmup 1 --graph off --no-cpu-baseline --no-roofline > /dev/null 2> $out/write.err
python3 $root/tools/pmc_traffic.py $out/fetch $out/write $out/${tag}_pmc_traffic.json > $out/${tag}_pmc_traffic.txt
rm -rf $out/fetch $out/write
echo "pmc done"
# 2b. MFMA utilisation of the matrix kernels (counter-only pass)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $out/mfma --output-format csv -- python3 $root/bench.py --steps 2 --warmup 1 --graph off --no-cpu-baseline --no-roofline > /dev/null 2> $out/mfma.err
python3 $root/tools/pmc_mfma.py $out/mfma > $out/${tag}_pmc_mfma.txt
rm -rf $out/mfma
echo "mfma done"
cd $root
# 3. the timeline of one captured step
tools/trace_step.sh
cp gpurun_out/gaps.txt $out/${tag}_step_kernels.txt
echo "trace done"
# 4. the bench lines themselves
python3 bench.py > $out/${tag}_bench_default.json 2> $out/default.err
python3 bench.py --model contextnet14 --steps 10 --warmup 3 --no-cpu-baseline > $out/${tag}_bench_contextnet14.json 2> $out/ctx.err
python3 bench.py --mode eval --steps 20 --warmup 3 > $out/${tag}_bench_eval_c5_fastscnn.json 2> $out/eval1.err
python3 bench.py --mode eval --steps 20 --warmup 3 --model contextnet14 > $out/${tag}_bench_eval_c5_contextnet14.json 2> $out/eval2.err
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --host-batch > /dev/null 2> $out/${tag}_host_batch.txt
echo "bench done"
