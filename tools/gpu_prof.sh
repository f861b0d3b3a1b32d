# rocprofv3 kernel trace + PMC passes of the benchmark (short run)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
ARGS="bench.py --steps 2 --warmup 1 --no-cpu --no-exclusive"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/trace -o t --output-format csv -- python3 $ARGS > gpurun_out/prof/trace.log 2>&1
echo trace done
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d gpurun_out/prof/pmc1 -o p --output-format csv -- python3 $ARGS > gpurun_out/prof/pmc1.log 2>&1
echo pmc1 done
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d gpurun_out/prof/pmc2 -o p --output-format csv -- python3 $ARGS > gpurun_out/prof/pmc2.log 2>&1
echo pmc2 done
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof/pmc3 -o p --output-format csv -- python3 $ARGS > gpurun_out/prof/pmc3.log 2>&1
echo pmc3 done
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU -d gpurun_out/prof/pmc4 -o p --output-format csv -- python3 $ARGS > gpurun_out/prof/pmc4.log 2>&1
echo pmc4 done
find gpurun_out/prof -name "*.csv" | head -30
