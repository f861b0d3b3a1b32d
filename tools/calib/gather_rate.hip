// Random 128-byte record gather rate (seven 16-byte loads per lane, as a BVH node is
// read) against the table size: L2-resident, Infinity-Cache-resident, HBM.
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void k_gather(const float4* table, uint32_t mask, float* out, uint32_t n, uint32_t rounds) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.f;
  uint32_t h = i * 2654435761u + 12345u;
  for (uint32_t r = 0; r < rounds; ++r) {
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    const float4* p = table + (size_t)(h & mask) * 8;
    float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4], f = p[5], g = p[6];
    acc += a.x + b.y + c.z + d.w + e.x + f.y + g.z;
    h += __float_as_uint(acc) & 1u;  // the next address depends on this load, as in a tree walk
  }
  if (i < n) out[i] = acc;
}

int main() {
  const uint32_t n = 256 * 20 * 64 * 4;  // 20 wavefronts per CU x 4
  float* out;
  hipMalloc(&out, n * 4);
  for (int lg = 14; lg <= 25; ++lg) {  // 2^lg records of 128 B: 2 MB .. 4 GB
    uint32_t recs = 1u << lg;
    float4* table;
    if (hipMalloc(&table, (size_t)recs * 128) != hipSuccess) break;
    hipMemset(table, 0, (size_t)recs * 128);
    const uint32_t rounds = 64;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k_gather, dim3(n / 64), dim3(64), 0, 0, table, recs - 1, out, n, rounds);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k_gather, dim3(n / 64), dim3(64), 0, 0, table, recs - 1, out, n, rounds);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    double bytes = (double)n * rounds * 128;
    printf("table %8.1f MB: %7.3f ms, %6.2f TB/s of 128-byte records (%.1f G records/s)\n", recs * 128.0 / 1e6, ms,
           bytes / ms / 1e9, n * (double)rounds / ms / 1e6);
    hipFree(table);
  }
  return 0;
}
