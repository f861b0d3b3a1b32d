# throughput of the splatting integrators (light tracing, BDPT) on the Cornell boxes at 1024x1024
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/splat.txt
while read name spp; do
timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --no-cpu --no-exclusive --scene $name --width 1024 --height 1024 --depth 8 --spp-per-step $spp > gpurun_out/s.json 2> gpurun_out/s.err || { echo "$name FAILED"; tail -5 gpurun_out/s.err; exit 1; }
python - <<'PY' | tee -a gpurun_out/splat.txt
import json
d = json.loads([l for l in open('gpurun_out/s.json') if l.startswith('{')][-1])
print('%-90s %8.1f Msamples/s %8.1f Mrays/s  spp/step %d' % (d['config']['workload'], d['value'], d['mrays_per_s'], d['config']['spp_per_step']), {k: round(v) for k, v in d['kernel_ms'].items()})
PY
done <<'LIST'
cornell_lt 48
cornell_fog_lt 48
cornell_bdpt 12
cornell_fog_bdpt 12
cornell_smoke_bdpt 12
LIST
# and on the benchmark scene (1.37 M primitives, 1920x1080)
for integ in lt bdpt; do
timeout -k 10 300 python -u bench.py --steps 2 --warmup 1 --no-cpu --no-exclusive --integrator $integ --spp-per-step 6 > gpurun_out/s.json 2> gpurun_out/s.err || { echo "car $integ FAILED"; tail -5 gpurun_out/s.err; exit 1; }
python - <<'PY' | tee -a gpurun_out/splat.txt
import json
d = json.loads([l for l in open('gpurun_out/s.json') if l.startswith('{')][-1])
print('%-90s %8.1f Msamples/s %8.1f Mrays/s  spp/step %d' % (d['config']['workload'], d['value'], d['mrays_per_s'], d['config']['spp_per_step']), {k: round(v) for k, v in d['kernel_ms'].items()})
PY
done
