# car_pt and stairs_pt with every build under variants/ plus the default one
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for sc in car_pt stairs_pt; do
for lib in cuda-pathtracer_amd/libcaster.so $(ls cuda-pathtracer_amd/variants/*.so 2>/dev/null) cuda-pathtracer_amd/libcaster.so; do
  CASTER_LIB=$PWD/$lib timeout -k 10 200 python -u bench.py --steps 4 --warmup 1 --no-cpu --no-exclusive --scene $sc > gpurun_out/v.json 2> gpurun_out/v.err || { echo "$lib FAILED"; tail -3 gpurun_out/v.err; continue; }
  python - "$lib" $sc <<'PY'
import json, sys
d = json.loads([l for l in open('gpurun_out/v.json') if l.startswith('{')][-1])
k = d['kernel_ms']
print('%-10s %-40s %7.1f Msamples/s  ext %.0f shade %.0f shadow %.0f probe %.0f' % (sys.argv[2], sys.argv[1][-40:], d['value'], k['extend'], k['shade'], k['shadow'], k['probe']))
PY
done
done
