set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python -u bench.py --steps 10 --warmup 2 > gpurun_out/bench_r2.json 2> gpurun_out/bench_r2.err || (tail -20 gpurun_out/bench_r2.err; exit 1)
python - <<'PY'
import json
d = json.load(open('gpurun_out/bench_r2.json'))
print(d['metric'], round(d['value'], 1), d['unit'], 'ms/step', round(d['ms_per_step'], 1))
print('cpu', d['cpu_baseline'])
r = d['roofline']
print({k: r[k] for k in r if k not in ('overlapped', 'exclusive')})
for k in r['exclusive'].get('kernels', []): print('excl', {a: (round(b, 4) if isinstance(b, float) else b) for a, b in k.items()})
for k in r['overlapped']['kernels']: print('ovl ', {a: (round(b, 4) if isinstance(b, float) else b) for a, b in k.items()})
print('pipeline', r['overlapped']['requested_rate_gbs'], r['exclusive'].get('requested_rate_gbs'))
PY
bash tools/gpu_prof.sh
