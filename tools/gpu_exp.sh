# runs one experiment script under tools/exp on the GPU box
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python -u tools/exp/$@ 2>&1 | tee gpurun_out/exp_$1.log
