// Random 128-byte record gather rate (seven 16-byte loads per lane, as a BVH node is
// read) against the table size: L2-resident, Infinity-Cache-resident, HBM.
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// LOADS 16-byte loads per record, records STRIDE float4s apart
template <int LOADS, int STRIDE>
__global__ void k_gather(const float4* table, uint32_t mask, float* out, uint32_t n, uint32_t rounds) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.f;
  uint32_t h = i * 2654435761u + 12345u;
  for (uint32_t r = 0; r < rounds; ++r) {
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    const float4* p = table + (size_t)(h & mask) * STRIDE;
    float4 v[LOADS];
#pragma unroll
    for (int k = 0; k < LOADS; ++k) v[k] = p[k];
#pragma unroll
    for (int k = 0; k < LOADS; ++k) acc += v[k].x + v[k].w;
    h += __float_as_uint(acc) & 1u;  // the next address depends on this load, as in a tree walk
  }
  if (i < n) out[i] = acc;
}

template <int LOADS, int STRIDE>
static void run(const char* what, float* out, uint32_t n) {
  printf("%s\n", what);
  for (int lg = 16; lg <= 24; lg += 2) {  // 2^lg records
    uint32_t recs = 1u << lg;
    float4* table;
    if (hipMalloc(&table, (size_t)recs * STRIDE * 16) != hipSuccess) break;
    hipMemset(table, 0, (size_t)recs * STRIDE * 16);
    const uint32_t rounds = 64;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_gather<LOADS, STRIDE>), dim3(n / 64), dim3(64), 0, 0, table, recs - 1, out, n, rounds);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((k_gather<LOADS, STRIDE>), dim3(n / 64), dim3(64), 0, 0, table, recs - 1, out, n, rounds);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    printf("  %8u records, table %8.1f MB: %7.3f ms, %6.1f G records/s, %6.1f G 16-byte lane loads/s, %6.2f TB/s loaded\n", recs,
           (double)recs * STRIDE * 16 / 1e6, ms, n * (double)rounds / ms / 1e6, n * (double)rounds * LOADS / ms / 1e6,
           (double)n * rounds * LOADS * 16 / ms / 1e9);
    hipFree(table);
  }
}

int main() {
  const uint32_t n = 256 * 20 * 64 * 4;  // 20 wavefronts per CU x 4
  float* out;
  hipMalloc(&out, n * 4);
  run<7, 8>("7 loads of a 128-byte record (a 4-wide node as the traversal reads it)", out, n);
  run<4, 4>("4 loads of a 64-byte record", out, n);
  run<3, 3>("3 loads of a 48-byte record (a primitive)", out, n);
  run<2, 2>("2 loads of a 32-byte record", out, n);
  run<1, 1>("1 load of a 16-byte record", out, n);
  return 0;
}
