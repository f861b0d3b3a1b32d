# PMC pass of the traversal kernel on the heterogeneous-medium Cornell box, two builds of the library
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/pmcab; mkdir -p gpurun_out/pmcab
ARGS="bench.py --steps 1 --warmup 1 --no-cpu --no-exclusive --scene cornell_smoke --width 1024 --height 1024 --depth 16"
for v in new old; do
  if [ $v = old ]; then export CASTER_LIB=$PWD/cuda-pathtracer_amd/variants/libcaster_old.so; else unset CASTER_LIB; fi
  timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD -d gpurun_out/pmcab/$v -o p --output-format csv -- python3 $ARGS > gpurun_out/pmcab/$v.log 2>&1 || { tail -5 gpurun_out/pmcab/$v.log; exit 1; }
  python3 tools/pmc_summary.py $(find gpurun_out/pmcab/$v -name "*counter_collection.csv") > gpurun_out/pmcab/$v.txt
  find gpurun_out/pmcab/$v -name "*.csv" -size +1M -delete
done
grep -A9 "^k_trace<false>" gpurun_out/pmcab/new.txt
grep -A9 "^k_trace<false>" gpurun_out/pmcab/old.txt
