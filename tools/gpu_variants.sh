# benchmark every build under cuda-pathtracer_amd/variants/ plus the default one
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/variants.txt
for lib in cuda-pathtracer_amd/libcaster.so $(ls cuda-pathtracer_amd/variants/*.so 2>/dev/null) cuda-pathtracer_amd/libcaster.so; do
  CASTER_LIB=$PWD/$lib timeout -k 10 200 python -u bench.py --steps 6 --warmup 2 --no-cpu > gpurun_out/v.json 2> gpurun_out/v.err || { echo "$lib FAILED"; tail -3 gpurun_out/v.err; continue; }
  python - "$lib" <<'PY' | tee -a gpurun_out/variants.txt
import json, sys
d = json.load(open('gpurun_out/v.json'))
k = d['kernel_ms']
ex = d.get('roofline', {}).get('exclusive', {})
exk = ' '.join('%s %.2f' % (e['kernel'][:14], e['avg_launch_ms']) for e in ex.get('kernels', []))
w = d.get('roofline', {}).get('gpu_walk', {})
print('%-45s %7.1f Msamples/s  ext %.1f shade %.1f shadow %.1f probe %.1f | one stream %.1f Msamples/s, avg launch ms: %s | nodes/ray %.2f prims/ray %.2f' % (
    sys.argv[1][-45:], d['value'], k['extend'], k['shade'], k['shadow'], k['probe'], ex.get('msamples_per_s', 0), exk, w.get('node_fetches_per_ray', 0), w.get('prim_tests_per_ray', 0)))
PY
done
