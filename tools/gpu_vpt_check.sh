# parity of the volumetric scenes + throughput of the volumetric configuration
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "vpt or smoke or golden or configs" > gpurun_out/vpt_tests.log 2>&1 || (tail -30 gpurun_out/vpt_tests.log; exit 1)
tail -2 gpurun_out/vpt_tests.log
for sc in cornell_vpt cornell_smoke; do
timeout -k 10 300 python -u bench.py --steps 3 --warmup 1 --no-cpu --no-exclusive --scene $sc --width 1024 --height 1024 --depth 16 > gpurun_out/c.json 2> gpurun_out/c.err || { tail -3 gpurun_out/c.err; exit 1; }
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/c.json') if l.startswith('{')][-1])
print('%-70s %8.1f Msamples/s %8.1f Mrays/s' % (d['config']['workload'][:70], d['value'], d['mrays_per_s']), {k: round(v) for k, v in d['kernel_ms'].items()})
PY
done
