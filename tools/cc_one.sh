#!/bin/bash
# compile ONE csrc/*.hip with build.py's flags and print the register budget / spills of every kernel in it (CPU container; no GPU needed)
#   tools/cc_one.sh pwsweep        [-S: also keep the ISA in /tmp/<name>.s]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/torch_semantic_segmentation_amd/csrc
n=$1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -I$R/include -I$C -c $C/$n.hip -o /tmp/$n.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|warning:|Function Name|VGPRs|ScratchSize|Occupancy" | sed -e 's/.*remark: *//' -e 's/\[-Rpass.*//'
if [ "$2" = "-S" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I$R/include -I$C -S --cuda-device-only $C/$n.hip -o /tmp/$n.s
  echo "ISA: /tmp/$n.s"
fi
