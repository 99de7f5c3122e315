#!/bin/bash
# Build variants of libtss_hip.so that differ in -D macros of ONE source file (here: prefetch depths of csrc/dwroll.hip) into
# torch_semantic_segmentation_amd/variants/ (git-ignored, travels with gpurun); run with TSS_HIP_LIB=<variant .so>.
#   tools/ab_variants.sh dwroll.hip name1 "-DTSS_ROLL_BPF1=3" name2 "-DTSS_ROLL_BPF1=6" ...
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
pkg=$root/torch_semantic_segmentation_amd
src=$1; shift
python -m torch_semantic_segmentation_amd.build > /dev/null
mkdir -p $pkg/variants
objs=$(ls $pkg/csrc/*.o | grep -v timing | grep -v "/${src%.*}.o")
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I$root/include -I$pkg/csrc $flags -c $pkg/csrc/$src -o $pkg/variants/$name.o \
    -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|Scratch" | sed 's/.*remark: *//' | paste - - - | sed "s/^/$name: /"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $pkg/variants/libtss_hip_$name.so $objs $pkg/variants/$name.o
  rm $pkg/variants/$name.o
done
ls $pkg/variants
