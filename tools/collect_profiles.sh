#!/bin/bash
# Everything profiles/ holds for one round, in one GPU call: tools/collect_profiles.sh r01
set -e
tag=${1:-r01}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/profiles_$tag
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# 1. per-kernel time of the bench command itself
rocprofv3 --kernel-trace --stats -d $out/stats --output-format csv -- python3 $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/${tag}_bench_under_rocprof.json 2> $out/stats.err
cp $out/stats/*/*kernel_stats.csv $out/${tag}_bench_kernel_stats.csv
cp $out/stats/*/*domain_stats.csv $out/${tag}_bench_domain_stats.csv
rm -rf $out/stats
echo "stats done"
# 2. HBM traffic counters: two counter-only passes over un-captured steps (one dispatch per kernel launch)
rocprofv3 --pmc FETCH_SIZE -d $out/fetch --output-format csv -- python3 $root/bench.py --steps 2 --warmup 1 --graph off --no-cpu-baseline --no-roofline --no-extras > /dev/null 2> $out/fetch.err
rocprofv3 --pmc WRITE_SIZE -d $out/write --output-format csv -- python3 $root/bench.py --steps 2 --warmup 1 --graph off --no-cpu-baseline --no-roofline --no-extras > /dev/null 2> $out/write.err
python3 $root/tools/pmc_traffic.py $out/fetch $out/write $out/${tag}_pmc_traffic.json > $out/${tag}_pmc_traffic.txt
rm -rf $out/fetch $out/write
echo "pmc done"
# 2b. MFMA utilisation of the matrix kernels (counter-only pass)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $out/mfma --output-format csv -- python3 $root/bench.py --steps 2 --warmup 1 --graph off --no-cpu-baseline --no-roofline --no-extras > /dev/null 2> $out/mfma.err
python3 $root/tools/pmc_mfma.py $out/mfma > $out/${tag}_pmc_mfma.txt
rm -rf $out/mfma
echo "mfma done"
# 2c. BASELINE config 5 with the ASPP head: per-kernel time of the eval forward, and MFMA busy of its matrix kernels
rocprofv3 --kernel-trace --stats -d $out/aspp --output-format csv -- python3 $root/bench.py --mode eval --model fastscnn_aspp --steps 20 --warmup 3 --no-cpu-baseline > $out/${tag}_bench_eval_c5_aspp_under_rocprof.json 2> $out/aspp.err
cp $out/aspp/*/*kernel_stats.csv $out/${tag}_eval_c5_aspp_kernel_stats.csv
rm -rf $out/aspp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $out/mfma2 --output-format csv -- python3 $root/bench.py --mode eval --model fastscnn_aspp --steps 2 --warmup 1 --graph off --no-cpu-baseline > /dev/null 2> $out/mfma2.err
python3 $root/tools/pmc_mfma.py $out/mfma2 > $out/${tag}_pmc_mfma_eval_c5_aspp.txt
rm -rf $out/mfma2
echo "aspp done"
cd $root
# 3. the timeline of one captured step
tools/trace_step.sh
cp gpurun_out/gaps.txt $out/${tag}_step_kernels.txt
cp gpurun_out/timeline.txt $out/${tag}_step_timeline.txt
tools/trace_step.sh --model contextnet14
cp gpurun_out/gaps.txt $out/${tag}_step_kernels_contextnet14.txt
TRACE_MARK=upsample_head tools/trace_step.sh --mode eval --model fastscnn_aspp
cp gpurun_out/gaps.txt $out/${tag}_eval_c5_aspp_kernels.txt
tools/trace_step.sh --model lednet
cp gpurun_out/gaps.txt $out/${tag}_step_kernels_lednet.txt
tools/trace_step.sh --model esnet
cp gpurun_out/gaps.txt $out/${tag}_step_kernels_esnet.txt
echo "trace done"
# 4. the bench lines themselves
python3 bench.py --host-batch > $out/${tag}_bench_default.json 2> $out/default.err     # the driver's command + the PCIe-inclusive legs; extras = configs 3 and 5
python3 bench.py --model contextnet14 --no-cpu-baseline --no-extras > $out/${tag}_bench_contextnet14.json 2> $out/ctx.err
python3 bench.py --mode eval --model fastscnn_aspp --steps 50 --warmup 5 > $out/${tag}_bench_eval_c5_aspp.json 2> $out/eval3.err
TSS_SYNCBN_FORCE=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --syncbn --no-cpu-baseline --no-extras --no-roofline 2> $out/syncbn.err | grep "^{" > $out/${tag}_bench_syncbn_1rank.json
TSS_SYNCBN_IPC=0 TSS_SYNCBN_FORCE=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 --syncbn --no-cpu-baseline --no-extras --no-roofline 2> $out/syncbn2.err | grep "^{" > $out/${tag}_bench_syncbn_1rank_allreduce.json
python3 tools/graph_memset_probe.py > $out/${tag}_memset_probe.log 2>&1; cp gpurun_out/memset_probe.txt $out/${tag}_memset_probe.txt
python3 tools/micro_atrous.py 2>&1 | grep dil > $out/${tag}_micro_atrous.txt
TSS_CONV3X3_WSTAT=0 python3 tools/micro_atrous.py 2>&1 | grep dil >> $out/${tag}_micro_atrous.txt
python3 bench.py --model lednet --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $out/${tag}_bench_lednet.json 2> $out/lednet.err
# "what you get without this project" (SURVEY 8d): the oracle's modules on the stock PyTorch-ROCm / MIOpen path, same batch, beside the headline
python3 bench.py --stock --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $out/${tag}_bench_stock.json 2> $out/stock.err
python3 tools/micro_sweep.py > $out/${tag}_micro_sweep.txt 2>&1
python3 tools/micro_fc1d.py 2>&1 | grep -v "amdgpu\|Warn" > $out/${tag}_micro_fc1d.txt
python3 tools/micro_sconv.py 2>&1 | grep -v "amdgpu\|Warn" > $out/${tag}_micro_sconv.txt
python3 bench.py --model esnet --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/${tag}_bench_esnet.json 2> $out/esnet.err
echo "bench done"
# 5. the parity table of the benchmarked kernels against the f64 oracle, written by the test itself: the tracked copy can not lag the code
python3 -m pytest tests/test_gpu_lean_vs_oracle.py -q -x > $out/lean_parity_pytest.log 2>&1 || true
cp gpurun_out/lean_parity.txt $out/${tag}_lean_parity.txt
tail -3 $out/lean_parity_pytest.log
echo "parity done"
# 6. where the waves of each kernel spend their cycles (one counter-only pass, 8 SQ counters)
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $out/wave --output-format csv -- python3 $root/bench.py --steps 2 --warmup 1 --graph off --no-cpu-baseline --no-roofline --no-extras > /dev/null 2> $out/wave.err
python3 $root/tools/pmc_wavestate.py $out/wave > $out/${tag}_pmc_wavestate.txt
rm -rf $out/wave
cd $root
echo "wavestate done"
