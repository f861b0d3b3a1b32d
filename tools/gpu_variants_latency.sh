# bench line and one-iteration latency of every build under cuda-pathtracer_amd/variants/ plus the default one
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/variants_latency.txt
for lib in cuda-pathtracer_amd/libcaster.so $(ls cuda-pathtracer_amd/variants/*.so 2>/dev/null); do
  echo "== $lib" | tee -a gpurun_out/variants_latency.txt
  CASTER_LIB=$PWD/$lib timeout -k 10 200 python -u bench.py --steps 5 --warmup 1 --no-cpu > gpurun_out/v.json 2> gpurun_out/v.err || { echo "$lib FAILED"; tail -3 gpurun_out/v.err; continue; }
  python - <<'PY' | tee -a gpurun_out/variants_latency.txt
import json
d = json.load(open('gpurun_out/v.json'))
k = d['kernel_ms']
print('%7.1f Msamples/s (one stream %7.1f)  ext %.0f shade %.0f shadow %.0f probe %.0f' % (d['value'], d['roofline']['exclusive']['msamples_per_s'], k['extend'], k['shade'], k['shadow'], k['probe']))
PY
  CASTER_LIB=$PWD/$lib timeout -k 10 200 python -u tools/exp/interactive_latency.py 2>/dev/null | grep -v "read-back" | tee -a gpurun_out/variants_latency.txt
done
