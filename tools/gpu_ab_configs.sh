# A/B of two builds over the configurations that do not use k_shade_pt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for lib in cuda-pathtracer_amd/libcaster.so cuda-pathtracer_amd/variants/contig.so; do
while read name w h depth integ; do
extra=""; [ -n "$integ" ] && extra="--integrator $integ"
CASTER_LIB=$PWD/$lib timeout -k 10 300 python -u bench.py --steps 3 --warmup 1 --no-cpu --no-exclusive --scene $name --width $w --height $h --depth $depth $extra > gpurun_out/c.json 2> gpurun_out/c.err || { echo "$name FAILED"; tail -3 gpurun_out/c.err; continue; }
python - "$lib" "$integ" <<'PY'
import json, sys
d = json.loads([l for l in open('gpurun_out/c.json') if l.startswith('{')][-1])
print('%-40s %-60s %8.1f Msamples/s' % (sys.argv[1][-40:], d['config']['workload'][:60] + ' ' + sys.argv[2], d['value']), {k: round(v) for k, v in d['kernel_ms'].items()})
PY
done <<'LIST'
stairs_pt 1920 1080 8
knot_pt 1024 1024 8
car_pt 1920 1080 8
LIST
done
