set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread > gpurun_out/check.log 2>&1 || (tail -60 gpurun_out/check.log; exit 1)
tail -3 gpurun_out/check.log
bash tools/gpu_configs_bench.sh
