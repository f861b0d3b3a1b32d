# 32 against 64 iterations per batch (three streams) on four scenes, twice each
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for sc in "stairs_pt 1920 1080 8" "knot_pt 1024 1024 8" "cornell_vpt 1024 1024 16" "car_pt 1920 1080 8"; do
set -- $sc
for it in 64 32 64 32; do
timeout -k 10 300 python -u bench.py --steps 4 --warmup 1 --no-cpu --no-exclusive --scene $1 --width $2 --height $3 --depth $4 --streams 3 --iters-in-flight $it --spp-per-step $((3*it)) > gpurun_out/p.json 2> gpurun_out/p.err || { echo "$1 $it FAILED"; tail -3 gpurun_out/p.err; continue; }
python -c "
import json
d = json.loads([l for l in open('gpurun_out/p.json') if l.startswith('{')][-1])
print('%-12s 3 x %-3s %8.1f Msamples/s' % ('$1', '$it', d['value']))"
done
done
