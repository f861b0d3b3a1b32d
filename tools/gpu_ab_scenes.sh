# libcaster.so and every build under variants/ on several configurations, same call:  SCENES="name w h depth [integrator];..." bash tools/gpu_ab_scenes.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
IFS=';' read -ra LIST <<< "${SCENES:-cornell_vpt 1024 1024 16;stairs_pt 1920 1080 8}"
for entry in "${LIST[@]}"; do
  set -- $entry
  extra=""; [ -n "$5" ] && extra="--integrator $5"
  for lib in cuda-pathtracer_amd/libcaster.so cuda-pathtracer_amd/variants/*.so cuda-pathtracer_amd/libcaster.so; do
    [ -f "$lib" ] || continue
    CASTER_LIB=$PWD/$lib timeout -k 10 300 python -u bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu --no-exclusive --scene $1 --width $2 --height $3 --depth $4 $extra > gpurun_out/v.json 2> gpurun_out/v.err || { echo "$lib FAILED"; tail -5 gpurun_out/v.err; exit 1; }
    python - "$1 $5" "$lib" <<'PY'
import json, sys
d = json.loads([l for l in open('gpurun_out/v.json') if l.startswith('{')][-1])
print('%-22s %-40s %8.1f Msamples/s' % (sys.argv[1], sys.argv[2][-40:], d['value']), {k: round(v) for k, v in d['kernel_ms'].items()})
PY
  done
done
