set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in ws_one ws_two; do
  echo "== $v"
  CASTER_LIB=$PWD/cuda-pathtracer_amd/variants/$v.so timeout -k 10 200 python -u tools/exp/walk_stats.py 2>&1 | tail -4
done | tee gpurun_out/walkstats.txt
