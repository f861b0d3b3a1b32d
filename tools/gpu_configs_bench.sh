# throughput of all five benchmark configurations (only the car scene is the bench line)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/configs.txt
while read name w h depth; do
timeout -k 10 300 python -u bench.py --steps 3 --warmup 1 --no-cpu --no-exclusive --scene $name --width $w --height $h --depth $depth > gpurun_out/c.json 2> gpurun_out/c.err || { echo "$name FAILED"; tail -3 gpurun_out/c.err; continue; }
python - <<'PY' | tee -a gpurun_out/configs.txt
import json
d = json.loads([l for l in open('gpurun_out/c.json') if l.startswith('{')][-1])
print('%-90s %8.1f Msamples/s %8.1f Mrays/s  spp/step %d' % (d['config']['workload'], d['value'], d['mrays_per_s'], d['config']['spp_per_step']), {k: round(v) for k, v in d['kernel_ms'].items()})
PY
done <<'LIST'
cornell_ao 512 512 0
knot_pt 1024 1024 8
stairs_pt 1920 1080 8
cornell_vpt 1024 1024 16
cornell_smoke 1024 1024 16
car_pt 1920 1080 8
LIST
