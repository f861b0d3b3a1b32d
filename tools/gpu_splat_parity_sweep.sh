# whole frames of light tracing and BDPT against the oracle (tools/exp/full_frame_splat_parity.py)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/splat_parity_sweep.txt
for job in "car_pt bdpt 2" "stairs_pt bdpt 1" "car_pt lt 6" "stairs_pt lt 3" "cornell_vpt bdpt 2 1024 1024" "cornell_vpt lt 6 1024 1024"; do
  set -- $job
  timeout -k 10 500 python -u tools/exp/full_frame_splat_parity.py $@ 2>&1 | grep -v "amdgpu.ids" | tee -a gpurun_out/splat_parity_sweep.txt | grep -v "oracle iteration" || true
done
