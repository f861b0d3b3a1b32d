set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread > gpurun_out/check.log 2>&1 || (tail -60 gpurun_out/check.log; exit 1)
tail -3 gpurun_out/check.log
bash tools/gpu_multi.sh
timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu > gpurun_out/b.json 2> gpurun_out/b.err || (tail gpurun_out/b.err; exit 1)
python -c "
import json; d=json.load(open('gpurun_out/b.json')); print('bench', round(d['value'],1), d['kernel_ms'])"
