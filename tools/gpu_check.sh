# parity tests then the benchmark
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 240 --timeout-method thread 2>&1 | tail -8
timeout -k 10 400 python -u bench.py --steps 5 --warmup 1 ${BENCH_ARGS:-} > gpurun_out/bench_latest.json 2> gpurun_out/bench_latest.err || (tail -20 gpurun_out/bench_latest.err; exit 1)
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_latest.json'))
print({k:d[k] for k in ('value','mrays_per_s','ms_per_step','invalid_samples','stack_overflow')})
print('kernel_ms', {k:round(v,1) for k,v in d['kernel_ms'].items()})
print('cpu', d.get('cpu_baseline',{}).get('value'), 'roof', d['roofline'].get('kernel'), d['roofline'].get('frac'))
PY
