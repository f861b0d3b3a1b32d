# throughput against scene size (same view, fewer triangles): how much of the time is cache misses
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for d in 1.0 0.25 0.05 0.01; do
timeout -k 10 300 python -u bench.py --steps 4 --warmup 1 --no-cpu --detail $d > gpurun_out/d.json 2> gpurun_out/d.err || { tail -3 gpurun_out/d.err; exit 1; }
python - $d <<'PY'
import json, sys
d = json.loads([l for l in open('gpurun_out/d.json') if l.startswith('{')][-1])
print('detail', sys.argv[1], d['config']['workload'].split(')')[0], '%.1f Msamples/s %.1f Mrays/s' % (d['value'], d['mrays_per_s']), {k: round(v) for k, v in d['kernel_ms'].items()})
PY
done
