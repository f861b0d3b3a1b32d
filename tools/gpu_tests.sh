# the whole GPU test suite, as the round driver runs it
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -v --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || (tail -40 gpurun_out/gpu_tests.log; exit 1)
tail -5 gpurun_out/gpu_tests.log
