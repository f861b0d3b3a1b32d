# bench line against streams x iterations per batch (pool sizes about the default 3 x 64)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/pool_sweep.txt
while read streams iters spp; do
timeout -k 10 300 python -u bench.py --steps 4 --warmup 1 --no-cpu --no-exclusive --streams $streams --iters-in-flight $iters --spp-per-step $spp > gpurun_out/p.json 2> gpurun_out/p.err || { echo "$streams x $iters FAILED"; tail -3 gpurun_out/p.err; continue; }
python - "$streams x $iters, $spp spp/step" <<'PY' | tee -a gpurun_out/pool_sweep.txt
import json, sys
d = json.loads([l for l in open('gpurun_out/p.json') if l.startswith('{')][-1])
print('%-28s %8.1f Msamples/s' % (sys.argv[1], d['value']), {k: round(v) for k, v in d['kernel_ms'].items()})
PY
done <<'LIST'
3 64 192
3 48 144
3 32 96
3 24 72
3 16 48
4 32 128
4 16 64
6 16 96
3 64 192
LIST
