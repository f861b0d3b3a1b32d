set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/gpu_multi.sh
timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu > gpurun_out/b.json 2> gpurun_out/b.err || (tail gpurun_out/b.err; exit 1)
python -c "
import json; d=json.load(open('gpurun_out/b.json')); print('bench', round(d['value'],1), d['config'], d['kernel_ms']); print(d['roofline'].get('exclusive',{}).get('msamples_per_s'))"
