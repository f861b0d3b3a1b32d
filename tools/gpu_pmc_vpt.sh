# PMC passes of the volumetric configuration (C4)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/pmcvpt
ARGS="bench.py --steps 1 --warmup 1 --no-cpu --no-exclusive --scene cornell_vpt --width 1024 --height 1024 --depth 16 --streams 1 --iters-in-flight 16 --spp-per-step 16"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d gpurun_out/pmcvpt/p1 -o p --output-format csv -- python3 $ARGS > gpurun_out/pmcvpt/p1.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_WR -d gpurun_out/pmcvpt/p2 -o p --output-format csv -- python3 $ARGS > gpurun_out/pmcvpt/p2.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/pmcvpt/t -o t --output-format csv -- python3 $ARGS > gpurun_out/pmcvpt/t.log 2>&1
echo done
