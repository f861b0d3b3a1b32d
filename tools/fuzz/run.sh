#!/bin/bash
# Builds the loader fuzzer with AddressSanitizer + UBSan (host code only) and runs it over seed files
# made by make_seeds.py.  CPU only.   usage: tools/fuzz/run.sh [rounds]
set -e
cd "$(dirname "$0")/../.."
out=${FUZZ_OUT:-/tmp/caster_fuzz}
mkdir -p $out/seeds
S=cuda-pathtracer_amd/csrc/scene
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -ffp-contract=off \
    -o $out/fuzz_loaders tools/fuzz/fuzz_loaders.cpp $S/mesh_io.cpp $S/scene.cpp $S/bvh.cpp -lz
python tools/fuzz/make_seeds.py $out/seeds
# scene files: generated beside their meshes in $out, where the mutations are written too
for sc in cornell_smoke car_pt; do
  python tools/gen_scenes.py $out $sc --width 32 --height 32 --detail 0.01 --env > /dev/null
  cp $out/scene.json $out/seeds/$sc.json
done
FUZZ_TMP=$out FUZZ_ROUNDS=${1:-2000} ASAN_OPTIONS=detect_leaks=0:allocator_may_return_null=1:max_allocation_size_mb=4096 $out/fuzz_loaders $out/seeds/*
