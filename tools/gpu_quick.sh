# parity tests of the traversal and the path tracer, whole frames, then a short bench with exclusive durations
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_whole_frames.py tests/test_zz_no_alloc.py -m gpu -x -q --timeout 400 --timeout-method thread > gpurun_out/quick.log 2>&1 || (tail -40 gpurun_out/quick.log; exit 1)
tail -2 gpurun_out/quick.log
timeout -k 10 400 python -u bench.py --steps 6 --warmup 2 --no-cpu > gpurun_out/bench_latest.json 2> gpurun_out/bench_latest.err || (tail -20 gpurun_out/bench_latest.err; exit 1)
python - <<'PY'
import json
d = json.load(open('gpurun_out/bench_latest.json'))
print('bench', round(d['value'], 1), 'one stream', d['roofline'].get('exclusive', {}).get('msamples_per_s'))
for k in d['roofline']['exclusive'].get('kernels', []):
    print('  excl', k['kernel'], 'ms/launch', round(k['avg_launch_ms'], 3), 'frac', round(k['frac'], 3))
print('kernel_ms', {k: round(v, 1) for k, v in d['kernel_ms'].items()})
PY
