set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -v -s --timeout 240 --timeout-method thread 2>&1 | tee gpurun_out/parity1.log | tail -60
