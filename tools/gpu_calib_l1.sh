# What TCP_TOTAL_CACHE_ACCESSES counts for gathers of 16-byte lane loads, and the rate the per-CU cache takes them at:
# tools/calib/gather_rate issues a known number of lane loads (1 310 720 lanes x 64 rounds x LOADS per launch).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/calib_l1 && mkdir -p gpurun_out/calib_l1
timeout -k 10 120 ./tools/calib/gather_rate > gpurun_out/calib_l1/rates.txt || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_RD -d gpurun_out/calib_l1/p1 -o p --output-format csv -- ./tools/calib/gather_rate > gpurun_out/calib_l1/p1.log 2>&1 || { tail -5 gpurun_out/calib_l1/p1.log; exit 1; }
python3 tools/pmc_summary.py $(find gpurun_out/calib_l1 -name "*counter_collection.csv") > gpurun_out/calib_l1/summary.txt
find gpurun_out/calib_l1 -name "*.csv" -size +1M -delete
cat gpurun_out/calib_l1/rates.txt gpurun_out/calib_l1/summary.txt
