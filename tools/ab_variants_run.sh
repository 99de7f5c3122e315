#!/bin/bash
# bench.py (100 steps, median) with every library variant of tools/ab_variants.sh, and the default build
for lib in default $(ls torch_semantic_segmentation_amd/variants/libtss_hip_*.so 2>/dev/null | grep -v timing); do
  if [ $lib = default ]; then unset TSS_HIP_LIB; else export TSS_HIP_LIB=$PWD/$lib; fi
  for model in fastscnn contextnet14; do
    r=$(python bench.py --model $model --no-cpu-baseline --no-extras --no-roofline --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['timing']['median_ms_per_step'], d['config']['final_loss'])")
    echo "$(basename $lib) $model $r"
  done
done
