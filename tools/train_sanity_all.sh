set -e
o=gpurun_out/train_sanity_r04.txt
echo "# tools/train_sanity.py <model> [big]: 300 captured bf16 train steps on one learnable batch (4 x 3 x 256 x 512; big: 8 x 3 x 512 x 1024), dropout active" > $o
for m in fastscnn contextnet14; do
  echo "## $m big, default (one-sweep 1x1 backward, ...)" >> $o; timeout -k 10 200 python tools/train_sanity.py $m big 2>&1 | grep "loss\|miou" >> $o
  echo "## $m big, TSS_PW_SWEEP=0 (round-3 operators)" >> $o; TSS_PW_SWEEP=0 timeout -k 10 200 python tools/train_sanity.py $m big 2>&1 | grep "loss\|miou" >> $o
done
echo "## lednet, default (fc1d / sconv / ssnbt kernels)" >> $o; timeout -k 10 200 python tools/train_sanity.py lednet 2>&1 | grep "loss\|miou" >> $o
echo "## lednet, TSS_FC1D=0 TSS_SCONV=0 TSS_SSNBT_TAIL=0 (generic kernels, four-operator tail)" >> $o; TSS_FC1D=0 TSS_SCONV=0 TSS_SSNBT_TAIL=0 timeout -k 10 300 python tools/train_sanity.py lednet 2>&1 | grep "loss\|miou" >> $o
echo "## esnet, default (fc1d / fcg / sconv kernels)" >> $o; timeout -k 10 200 python tools/train_sanity.py esnet 2>&1 | grep "loss\|miou" >> $o
echo "## esnet, TSS_FC1D=0 TSS_FCG=0 TSS_SCONV=0 (generic kernels)" >> $o; TSS_FC1D=0 TSS_FCG=0 TSS_SCONV=0 timeout -k 10 300 python tools/train_sanity.py esnet 2>&1 | grep "loss\|miou" >> $o
cat $o
