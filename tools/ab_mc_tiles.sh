cd $GRAFT_REPO_ROOT
for t in 192 600 1100 100000 192; do
  TSS_PW_MC_SMALL=$t python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/ab.json 2>/dev/null
  echo "mc_thr32=$t $(grep -o 'ms_per_step.: [0-9.]*' gpurun_out/ab.json)"
done
