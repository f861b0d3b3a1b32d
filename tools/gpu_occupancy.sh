# traversal wavefronts per CU against launch durations with the GPU to itself (one stream)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in 20 16 14 12 8; do
  CASTER_WALK_PER_CU=$w timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/v.json 2> gpurun_out/v.err || { echo "$w FAILED"; tail -3 gpurun_out/v.err; exit 1; }
  python - $w <<'PY'
import json, sys
d = json.load(open('gpurun_out/v.json'))
ex = d['roofline']['exclusive']
print('walk per CU %-3s three streams %7.1f  one stream %7.1f  %s' % (sys.argv[1], d['value'], ex['msamples_per_s'], ' '.join('%s %.2f' % (e['kernel'][:14], e['avg_launch_ms']) for e in ex['kernels'])))
PY
done
