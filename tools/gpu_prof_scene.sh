# rocprofv3 kernel trace (three streams and one) + PMC passes of bench.py on another of the benchmark configurations:
#   [EXTRA_ARGS="--integrator bdpt"] bash tools/gpu_prof_scene.sh NAME SCENE WIDTH HEIGHT DEPTH      -> gpurun_out/prof_NAME/{kernel_stats.csv,kernel_stats_one_stream.csv,pmc_summary.txt}
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
NAME=$1
OUT=gpurun_out/prof_$NAME
rm -rf $OUT && mkdir -p $OUT
ARGS="bench.py --steps 2 --warmup 1 --no-cpu --no-exclusive --scene $2 --width $3 --height $4 --depth $5 ${EXTRA_ARGS:-}"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python3 $ARGS > $OUT/trace.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace1 -o t --output-format csv -- python3 $ARGS --streams 1 > $OUT/trace1.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d $OUT/pmc1 -o p --output-format csv -- python3 $ARGS > $OUT/pmc1.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d $OUT/pmc2 -o p --output-format csv -- python3 $ARGS > $OUT/pmc2.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc3 -o p --output-format csv -- python3 $ARGS > $OUT/pmc3.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU -d $OUT/pmc4 -o p --output-format csv -- python3 $ARGS > $OUT/pmc4.log 2>&1
python3 tools/pmc_summary.py --json $OUT/pmc.json $(find $OUT/pmc* -name "*counter_collection.csv") > $OUT/pmc_summary.txt
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
cp $(find $OUT/trace1 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_one_stream.csv
find $OUT -name "*.csv" -size +1M -delete
grep -h "Msamples" $OUT/trace.log | tail -1 | cut -c1-200 || true
head -12 $OUT/kernel_stats_one_stream.csv | cut -c1-160
