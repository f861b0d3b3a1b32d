# PMC pass on the scalar side of the benchmark kernels (scalar and branch instructions, instruction fetch), one stream
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/pmcsc; mkdir -p gpurun_out/pmcsc
ARGS="bench.py --steps 2 --warmup 1 --no-cpu --no-exclusive --streams 1 --iters-in-flight 8 --spp-per-step 8"
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES -d gpurun_out/pmcsc/p1 -o p --output-format csv -- python3 $ARGS > gpurun_out/pmcsc/p1.log 2>&1 || { tail -5 gpurun_out/pmcsc/p1.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_IFETCH SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d gpurun_out/pmcsc/p2 -o p --output-format csv -- python3 $ARGS > gpurun_out/pmcsc/p2.log 2>&1 || { tail -5 gpurun_out/pmcsc/p2.log; exit 1; }
python3 tools/pmc_summary.py $(find gpurun_out/pmcsc -name "*counter_collection.csv") > gpurun_out/pmcsc/summary.txt
find gpurun_out/pmcsc -name "*.csv" -size +1M -delete
grep -A18 "^k_trace<false>" gpurun_out/pmcsc/summary.txt
