set -e
cd $GRAFT_REPO_ROOT
TSS_TIMING=1 python -m torch_semantic_segmentation_amd.build > /dev/null
for cfg in "96 576 16384" "576 96 16384" "128 768 16384" "768 128 16384" "64 384 65536" "128 128 262144"; do
  echo "== wgrad K N P = $cfg"
  TSS_TIMING=1 python tools/micro_one.py wgrad $cfg
done
