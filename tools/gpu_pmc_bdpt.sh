# kernel trace and two PMC passes of the bidirectional integrator (Cornell box 1024x1024)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/bdptprof
ARGS="bench.py --steps 1 --warmup 1 --no-cpu --no-exclusive --scene ${1:-cornell_bdpt} --width 1024 --height 1024 --spp-per-step 6"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/bdptprof/trace -o t --output-format csv -- python3 $ARGS > gpurun_out/bdptprof/trace.log 2>&1 || exit 1
echo trace done
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d gpurun_out/bdptprof/pmc1 -o p --output-format csv -- python3 $ARGS > gpurun_out/bdptprof/pmc1.log 2>&1 || exit 1
echo pmc1 done
timeout -s KILL 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM -d gpurun_out/bdptprof/pmc2 -o p --output-format csv -- python3 $ARGS > gpurun_out/bdptprof/pmc2.log 2>&1 || exit 1
echo pmc2 done
