# rehearsal of the N>1 bench path on ONE GPU: bench.py starts its own two ranks, both on device 0, films meet over gloo
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for shard in spp tile; do
timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu --no-exclusive --backend gloo --share-gpu --shard $shard > gpurun_out/multi_$shard.json 2> gpurun_out/multi_$shard.err || (tail -20 gpurun_out/multi_$shard.err; exit 1)
python -c "
import json; d=json.loads([l for l in open('gpurun_out/multi_$shard.json') if l.startswith('{')][-1]); print('$shard', d['n_gpus'], round(d['value'],1), d['scaling'], d['config']['sharding'], d['rays'])"
done
