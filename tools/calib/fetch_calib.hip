// Calibrates rocprofv3's FETCH_SIZE for the traversal kernel's access pattern:
// every lane gathers one 128-byte record (seven 16-byte loads, as a BVH node is
// read) from a table far larger than the 256 MiB Infinity Cache, at a
// pseudo-random index.  The bytes that must come from HBM are known
// (n_gathers x 128), so FETCH_SIZE / known gives the counter's scale for this
// pattern.  Build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void k_gather(const float4* table, uint32_t n_records, float* out, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t h = i * 2654435761u;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  const float4* r = table + (size_t)(h % n_records) * 8;
  float4 a = r[0], b = r[1], c = r[2], d = r[3], e = r[4], f = r[5], g = r[6];
  out[i] = a.x + b.y + c.z + d.w + e.x + f.y + g.z;
}

__global__ void k_stream(const float4* table, size_t n_f4, float* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  float s = 0.f;
  for (; i < n_f4; i += (size_t)gridDim.x * blockDim.x) { float4 v = table[i]; s += v.x + v.w; }
  if (s == 123.456f) out[0] = s;
}

int main() {
  const uint32_t n_records = 48u << 20;            // 6 GiB of 128-byte records
  const uint32_t n = 32u << 20;                    // gathers: 4 GiB of distinct-line traffic at most
  float4* table; float* out;
  if (hipMalloc(&table, (size_t)n_records * 128) != hipSuccess) return 1;
  if (hipMalloc(&out, (size_t)n * 4) != hipSuccess) return 1;
  hipMemset(table, 0, (size_t)n_records * 128);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(k_gather, dim3(n / 256), dim3(256), 0, 0, table, n_records, out, n);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(k_stream, dim3(256 * 16), dim3(256), 0, 0, table, (size_t)n_records * 8, out);
  hipDeviceSynchronize();
  printf("gather: %u records x 128 B = %.3f GB expected (indices repeat, but the 6 GiB table is 24x the Infinity Cache: ~4% of repeats can hit)\n", n,
         n * 128.0 / 1e9);
  printf("stream: %.3f GB expected\n", n_records * 128.0 / 1e9);
  hipFree(table); hipFree(out);
  return 0;
}
