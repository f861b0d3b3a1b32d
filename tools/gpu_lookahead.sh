# the look-ahead window: its test, then the latency of one-iteration calls with and without it
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_golden.py -k "lookahead" -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/look.log 2>&1 || (tail -40 gpurun_out/look.log; exit 1)
tail -2 gpurun_out/look.log
timeout -k 10 400 python -u tools/exp/interactive_latency.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/latency.txt
