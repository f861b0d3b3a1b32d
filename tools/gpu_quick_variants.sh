set -e
bash tools/gpu_quick.sh
bash tools/gpu_variants.sh
