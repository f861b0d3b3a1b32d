# rocprofv3 kernel trace + PMC passes of the benchmark (short run), each counter set in a run of its own
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/prof
mkdir -p gpurun_out/prof
ARGS="bench.py --steps 2 --warmup 1 --no-cpu --no-exclusive"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/trace -o t --output-format csv -- python3 $ARGS > gpurun_out/prof/trace.log 2>&1
echo trace done
# the same on one stream: every launch has the GPU to itself (the durations bench.py's `exclusive` block reports)
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/trace1 -o t --output-format csv -- python3 $ARGS --streams 1 > gpurun_out/prof/trace1.log 2>&1
echo trace1 done
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d gpurun_out/prof/pmc1 -o p --output-format csv -- python3 $ARGS > gpurun_out/prof/pmc1.log 2>&1
echo pmc1 done
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d gpurun_out/prof/pmc2 -o p --output-format csv -- python3 $ARGS > gpurun_out/prof/pmc2.log 2>&1
echo pmc2 done
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof/pmc3 -o p --output-format csv -- python3 $ARGS > gpurun_out/prof/pmc3.log 2>&1
echo pmc3 done
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU -d gpurun_out/prof/pmc4 -o p --output-format csv -- python3 $ARGS > gpurun_out/prof/pmc4.log 2>&1
echo pmc4 done
# summaries (the raw per-dispatch CSVs are tens of MB)
python3 tools/pmc_summary.py --json gpurun_out/prof/pmc.json $(find gpurun_out/prof/pmc* -name "*counter_collection.csv") > gpurun_out/prof/pmc_summary.txt
cp $(find gpurun_out/prof/trace -name "*kernel_stats.csv" | head -1) gpurun_out/prof/kernel_stats.csv
cp $(find gpurun_out/prof/trace1 -name "*kernel_stats.csv" | head -1) gpurun_out/prof/kernel_stats_one_stream.csv
find gpurun_out/prof -name "*.csv" -size +1M -delete
find gpurun_out/prof -name "*.csv" | head -30
