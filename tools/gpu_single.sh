# one-stream run of the benchmark: per-kernel-class times without overlap between streams
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -u bench.py --steps 4 --warmup 1 --no-cpu --streams 1 --iters-in-flight 32 --spp-per-step 32 > gpurun_out/single.json 2> gpurun_out/single.err || { tail -5 gpurun_out/single.err; exit 1; }
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/single.json') if l.startswith('{')][-1])
k = d['kernel_ms']; tot = sum(k.values())
print('%.1f Msamples/s, %.1f ms/step; kernel classes (ms, %% of their sum %.1f ms of %.1f ms wall):' % (d['value'], d['ms_per_step'], tot, d['ms_per_step'] * d['steps']))
for n, v in k.items(): print('  %-9s %8.1f  %5.1f%%' % (n, v, 100 * v / tot))
PY
