# parity of the path tracer's shade stage first, then the variants bench
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_whole_frames.py tests/test_gpu_configs.py -m gpu -x -q --timeout 400 --timeout-method thread > gpurun_out/check.log 2>&1 || (tail -60 gpurun_out/check.log; exit 1)
tail -3 gpurun_out/check.log
bash tools/gpu_variants.sh
