# bench.py on libcaster.so and on every build under cuda-pathtracer_amd/variants/ (tuning experiments), same call
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for lib in cuda-pathtracer_amd/libcaster.so cuda-pathtracer_amd/variants/*.so; do
  [ -f "$lib" ] || continue
  CASTER_LIB=$PWD/$lib timeout -k 10 300 python -u bench.py --steps ${STEPS:-5} --warmup 1 --no-cpu ${BENCH_ARGS:-} > gpurun_out/v.json 2> gpurun_out/v.err || { echo "$lib FAILED"; tail -5 gpurun_out/v.err; exit 1; }
  python - "$lib" <<'PY'
import json, sys
d = json.loads([l for l in open('gpurun_out/v.json') if l.startswith('{')][-1])
ex = d.get('roofline', {}).get('exclusive', {}).get('kernels', [])
print('%-46s %8.1f Msamples/s' % (sys.argv[1][-46:], d['value']), {k: round(v) for k, v in d['kernel_ms'].items()},
      'alone ms:', {k['kernel'][:14]: round(k.get('avg_ms', 0), 3) for k in ex if isinstance(k, dict) and 'kernel' in k})
PY
done
