# the present build against an older one (cuda-pathtracer_amd/variants/libcaster_old.so), scene by scene, twice each
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for lib in cuda-pathtracer_amd/libcaster.so cuda-pathtracer_amd/variants/libcaster_old.so cuda-pathtracer_amd/libcaster.so cuda-pathtracer_amd/variants/libcaster_old.so; do
for sc in "cornell_vpt 1024 1024 16" "cornell_ao 512 512 0" "knot_pt 1024 1024 8" "car_pt 1920 1080 8"; do
set -- $sc
CASTER_LIB=$PWD/$lib timeout -k 10 300 python -u bench.py --steps 4 --warmup 1 --no-cpu --scene $1 --width $2 --height $3 --depth $4 > gpurun_out/c.json 2> gpurun_out/c.err || { echo FAILED; tail -3 gpurun_out/c.err; exit 1; }
python -c "
import json
d = json.loads([l for l in open('gpurun_out/c.json') if l.startswith('{')][-1])
ex = d['roofline'].get('exclusive', {})
print('$lib'[-20:], '%-12s' % '$1', round(d['value'],1), 'one stream', round(ex.get('msamples_per_s', 0), 1), ' '.join('%s %.2f' % (e['kernel'][:14], e['avg_launch_ms']) for e in ex.get('kernels', [])))"
done
done
