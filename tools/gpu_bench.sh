# first benchmark + kernel-trace profile on the GPU box
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 120 python -u __graft_entry__.py smoke 2>&1 | tail -3
timeout -k 10 400 python -u bench.py --steps 5 --warmup 1 > gpurun_out/bench1.json 2> gpurun_out/bench1.err || (tail -20 gpurun_out/bench1.err; exit 1)
cat gpurun_out/bench1.json
