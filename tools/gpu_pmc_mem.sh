# PMC passes on the vector-memory pipeline (address unit, L1) of the benchmark kernels
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/pmcmem
ARGS="bench.py --steps 2 --warmup 1 --no-cpu --streams 1 --iters-in-flight 8 --spp-per-step 8"
timeout -s KILL 200 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY SQ_BUSY_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAVE_CYCLES -d gpurun_out/pmcmem/p1 -o p --output-format csv -- python3 $ARGS > gpurun_out/pmcmem/p1.log 2>&1
echo p1 done
timeout -s KILL 200 rocprofv3 --pmc TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE -d gpurun_out/pmcmem/p2 -o p --output-format csv -- python3 $ARGS > gpurun_out/pmcmem/p2.log 2>&1
echo p2 done
