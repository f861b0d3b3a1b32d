# whole 1080p frames against the oracle at more samples than the tests take (tools/exp/full_frame_parity.py)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/parity_sweep.txt
for job in "car_pt 48 1.0" "stairs_pt 48 1.0" "knot_pt 32 1.0" "cornell_pt 32 1.0" "cornell_vpt 24 1.0" "cornell_smoke 12 1.0"; do
  set -- $job
  timeout -k 10 500 python -u tools/exp/full_frame_parity.py $1 $2 $3 16 2>&1 | grep -v "amdgpu.ids\|oracle rows" | tee -a gpurun_out/parity_sweep.txt
done
