# volumetric configuration with every build under variants/ plus the default one
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for lib in cuda-pathtracer_amd/libcaster.so $(ls cuda-pathtracer_amd/variants/*.so 2>/dev/null); do
  CASTER_LIB=$PWD/$lib timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --no-cpu --no-exclusive --scene cornell_vpt --width 1024 --height 1024 --depth 16 > gpurun_out/v.json 2> gpurun_out/v.err || { echo "$lib FAILED"; tail -3 gpurun_out/v.err; continue; }
  python - "$lib" <<'PY'
import json, sys
d = json.loads([l for l in open('gpurun_out/v.json') if l.startswith('{')][-1])
print('%-45s %7.1f Msamples/s  shade %.0f ms' % (sys.argv[1][-45:], d['value'], d['kernel_ms']['shade']))
PY
done
