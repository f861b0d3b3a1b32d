# resident wavefronts per CU of the traversal and the shade kernels, three streams: do they run side by side?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/cosched.txt
for cfg in "20 99" "16 99" "12 99" "16 8" "12 8" "16 12" "20 8" "20 99"; do
  set -- $cfg
  CASTER_WALK_PER_CU=$1 CASTER_SHADE_PER_CU=$2 timeout -k 10 200 python -u bench.py --steps 6 --warmup 2 --no-cpu --no-exclusive > gpurun_out/v.json 2> gpurun_out/v.err || { echo "$cfg FAILED"; tail -3 gpurun_out/v.err; exit 1; }
  python - "$cfg" <<'PY' | tee -a gpurun_out/cosched.txt
import json, sys
d = json.loads([l for l in open('gpurun_out/v.json') if l.startswith('{')][-1])
k = d['kernel_ms']
print('walk/shade per CU %-8s %7.1f Msamples/s  ext %.0f shade %.0f shadow %.0f' % (sys.argv[1], d['value'], k['extend'], k['shade'], k['shadow']))
PY
done
