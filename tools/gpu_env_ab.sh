# bench.py under several settings of the library's environment knobs and bench arguments, same call, twice over:
#   VARS="A=1 B=2;--streams 4|C=3;|;--streams 2" bash tools/gpu_env_ab.sh        (entries "ENV SETTINGS;BENCH ARGUMENTS")
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
IFS='|' read -ra SETS <<< "${VARS:-;}"
for rep in 1 2; do
for set in "${SETS[@]}" ";"; do
  envs="${set%%;*}"; args="${set#*;}"
  env $envs timeout -k 10 300 python -u bench.py --steps ${STEPS:-5} --warmup 1 --no-cpu --no-exclusive ${BENCH_ARGS:-} $args > gpurun_out/v.json 2> gpurun_out/v.err || { echo "$set FAILED"; tail -5 gpurun_out/v.err; exit 1; }
  python - "$set" <<'PY'
import json, sys
d = json.loads([l for l in open('gpurun_out/v.json') if l.startswith('{')][-1])
print('%-44s %8.1f Msamples/s  spp/step %d' % (sys.argv[1], d['value'], d['config']['spp_per_step']), {k: round(v) for k, v in d['kernel_ms'].items()})
PY
done
done
