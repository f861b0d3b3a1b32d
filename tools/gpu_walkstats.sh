# lanes per step kind and the traversal stack's behaviour, from builds made with -DCASTER_WALK_STATS=1 under variants/
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in cuda-pathtracer_amd/variants/ws_*.so; do
  echo "== $v"
  CASTER_LIB=$PWD/$v timeout -k 10 200 python -u tools/exp/walk_stats.py 2>&1 | tail -8
done | tee gpurun_out/walkstats.txt
