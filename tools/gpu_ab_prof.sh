cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for lib in libcaster.so variants/r2kernels.so; do
  export CASTER_LIB=$PWD/cuda-pathtracer_amd/$lib
  tag=$(basename $lib .so)
  rm -rf gpurun_out/abp_$tag; mkdir -p gpurun_out/abp_$tag
  timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d gpurun_out/abp_$tag/p -o p --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-exclusive --scene stairs_pt --streams 1 > gpurun_out/abp_$tag/log 2>&1 || exit 1
  python3 tools/pmc_summary.py $(find gpurun_out/abp_$tag -name "*counter_collection.csv") > gpurun_out/abp_$tag/summary.txt
  find gpurun_out/abp_$tag -name "*.csv" -size +1M -delete
  echo $tag; grep -A12 "^k_trace<true>" gpurun_out/abp_$tag/summary.txt
done
