# bench line of the always / never builds of the probe-in-shadow path at several batch sizes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/probe_any_sweep.txt
for k in 1 2 4 8 16 64; do
  for lib in never always; do
    CASTER_LIB=$PWD/cuda-pathtracer_amd/variants/$lib.so timeout -k 10 200 python -u bench.py --steps 4 --warmup 1 --no-cpu --no-exclusive --iters-in-flight $k --spp-per-step $((k*3 < 24 ? 24 : k*3)) > gpurun_out/v.json 2> gpurun_out/v.err || { echo "$lib $k FAILED"; tail -3 gpurun_out/v.err; exit 1; }
    python -c "
import json; d=json.load(open('gpurun_out/v.json')); print('k=%-3d %-7s %8.1f Msamples/s' % ($k, '$lib', d['value']))" | tee -a gpurun_out/probe_any_sweep.txt
  done
done
