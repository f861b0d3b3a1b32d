# instruction-cache counters of the benchmark kernels (one stream)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/pmcic
ARGS="bench.py --steps 2 --warmup 1 --no-cpu --streams 1 --iters-in-flight 8 --spp-per-step 8"
timeout -s KILL 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU -d gpurun_out/pmcic/p1 -o p --output-format csv -- python3 $ARGS > gpurun_out/pmcic/p1.log 2>&1
echo p1 done
