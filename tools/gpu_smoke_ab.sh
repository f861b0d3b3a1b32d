# the density-grid tests, then bench.py on the smoke scene with libcaster.so and with every build under variants/ (same call, twice over)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 600 python -u -m pytest tests/test_gpu_whole_frames.py tests/test_gpu_golden.py -m gpu -x -q --timeout 300 --timeout-method thread -k "smoke or density or hetero or grid" > gpurun_out/smoke_tests.log 2>&1 || { tail -30 gpurun_out/smoke_tests.log; exit 1; }
tail -3 gpurun_out/smoke_tests.log
fi
for rep in 1 2; do
for lib in cuda-pathtracer_amd/libcaster.so cuda-pathtracer_amd/variants/*.so; do
  [ -f "$lib" ] || continue
  CASTER_LIB=$PWD/$lib timeout -k 10 300 python -u bench.py --steps ${STEPS:-6} --warmup 1 --no-cpu --no-exclusive --scene ${SCENE:-cornell_smoke} --width 1024 --height 1024 --depth 16 > gpurun_out/v.json 2> gpurun_out/v.err || { echo "$lib FAILED"; tail -5 gpurun_out/v.err; exit 1; }
  python - "$lib" <<'PY'
import json, sys
d = json.loads([l for l in open('gpurun_out/v.json') if l.startswith('{')][-1])
print('%-46s %8.1f Msamples/s' % (sys.argv[1][-46:], d['value']), {k: round(v) for k, v in d['kernel_ms'].items()})
PY
done
done
