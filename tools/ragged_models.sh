# whole LEDNet / ESNet train steps at a size whose maps are narrower than the kernels' tiles (2 x 3 x 192 x 320), lean kernels on and off: step time and final loss
for m in lednet esnet; do
  for env in "" "TSS_FC1D=0 TSS_FCG=0 TSS_SCONV=0 TSS_SSNBT_TAIL=0"; do
    echo "== $m [$env]"
    env $env timeout -k 10 200 python bench.py --model $m --batch 2 --height 192 --width 320 --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['config'].get('final_loss'))"
  done
done
