# Rehearsals of the N > 1 path on ONE GPU (every rank on device 0, films meeting over gloo): configuration 3 with
# tiles over 2 and 4 ranks, the benchmark scene over 8 ranks in both shard modes.  Unmeasured on real peers.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {  # name, args...
  name=$1; shift
  timeout -k 10 400 python -u bench.py "$@" --share-gpu --backend gloo --no-cpu > gpurun_out/multi_$name.json 2> gpurun_out/multi_$name.err || { echo "$name FAILED"; tail -5 gpurun_out/multi_$name.err; return 1; }
  python - $name <<'PY'
import json, sys
d = json.loads([l for l in open('gpurun_out/multi_%s.json' % sys.argv[1]) if l.startswith('{')][-1])
g = d.get('process_group') or {}
print('%-14s n_gpus %d %-5s %8.1f Msamples/s  %s  backend %s world %s devices %s' % (
    sys.argv[1], d['n_gpus'], d['scaling'], d['value'], d['config']['workload'][:40], g.get('backend'), g.get('world_size'),
    sorted(set((r['device_index'], r['device']) for r in g.get('ranks', [])))))
PY
}
run stairs_1 --scene stairs_pt --gpus 1 --steps 3 --warmup 1 --no-exclusive &&
run stairs_tile2 --scene stairs_pt --gpus 2 --shard tile --steps 3 --warmup 1 &&
run stairs_tile4 --scene stairs_pt --gpus 4 --shard tile --steps 3 --warmup 1 &&
run car_spp8 --gpus 8 --steps 2 --warmup 1 &&
run car_tile8 --gpus 8 --shard tile --steps 2 --warmup 1
