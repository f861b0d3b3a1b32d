# first GPU call of the round: the new entry point's test, the whole-frame tests, then the whole suite
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/wip_no_alloc.py -m gpu -x -v --timeout 200 --timeout-method thread > gpurun_out/noalloc.log 2>&1 || (tail -40 gpurun_out/noalloc.log; exit 1)
tail -3 gpurun_out/noalloc.log
timeout -k 10 600 python -u -m pytest tests/test_gpu_whole_frames.py -m gpu -x -v -s --timeout 400 --timeout-method thread > gpurun_out/whole.log 2>&1 || (tail -60 gpurun_out/whole.log; exit 1)
tail -5 gpurun_out/whole.log
