# Calibrates the L1 -> L2 request counter for the two access patterns of the renderer (streaming 16 B per lane; gathers of
# one 128-byte record per lane): tools/calib/fetch_calib moves a known number of bytes with each.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/calib && mkdir -p gpurun_out/calib
timeout -s KILL 120 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_READ_sum -d gpurun_out/calib/p1 -o p --output-format csv -- ./tools/calib/fetch_calib > gpurun_out/calib/p1.log 2>&1 || { tail -5 gpurun_out/calib/p1.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/calib/p2 -o p --output-format csv -- ./tools/calib/fetch_calib > gpurun_out/calib/p2.log 2>&1 || { tail -5 gpurun_out/calib/p2.log; exit 1; }
python3 tools/pmc_summary.py $(find gpurun_out/calib -name "*counter_collection.csv") > gpurun_out/calib/summary.txt
cat gpurun_out/calib/p1.log | grep -E "gather|stream"
cat gpurun_out/calib/summary.txt
