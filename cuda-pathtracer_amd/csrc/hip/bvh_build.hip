// Binned-SAH BVH build on the GPU, producing bit for bit the tree and the
// primitive order of the host build (scene/bvh.cpp, which restates
// CUDA-PathTracer/tracer/bvh.cpp:8-180): 12 buckets on the centroid along all
// three axes, a leaf at <= 4 primitives or an extent below 1e-4 or when no split
// beats N * area(parent); nodes numbered in pre-order, primitives leaf by leaf.
//
// The host build recurses node by node; here a whole level of the tree is
// processed at once.  `order` holds the primitive permutation, every node owns a
// contiguous segment of it, and per level:
//   classify   leaf by size / extent                               (per node)
//   bin        count and bound the primitives of every bucket      (per primitive,
//              LDS-privatised for the node the workgroup starts in, then atomics)
//   decide     evaluate the 3 x 11 candidate planes exactly as the host does,
//              child boxes = unions of bucket boxes                 (per node)
//   partition  stable partition of every split segment: one device-wide
//              exclusive scan of the "goes left" flags serves all segments
// Minimum and maximum are exact and order independent, so the bucket boxes equal
// the host's; the plane costs are evaluated by one thread in the host's order.
// A last pass turns level order into the host's pre-order numbering.
#include <hip/hip_runtime.h>

#include <string.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include <rocprim/device/device_scan.hpp>

#include "../scene/scene.h"

namespace caster {
namespace gpubvh {

constexpr int kBuckets = 12;
constexpr int kBinInts = 3 * kBuckets * 7;  // per node: 3 axes x 12 buckets x (count, lo.xyz, hi.xyz)
constexpr int kBlock = 256;

#define BVHCHECK(expr)                                                        \
  do {                                                                        \
    hipError_t e_ = (expr);                                                   \
    if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); return false; } \
  } while (0)

// order-preserving map float -> uint32 (for atomicMin / atomicMax on floats)
__host__ __device__ inline uint32_t f2key(float f) {
  uint32_t b;
#if defined(__HIP_DEVICE_COMPILE__)
  b = __float_as_uint(f);
#else
  memcpy(&b, &f, 4);
#endif
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ inline float key2f(uint32_t k) {
  uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(b);
}

struct NodeArrays {
  float* lo;    // 3 per node
  float* hi;    // 3 per node
  int* begin;   // segment [begin, end) of `order`
  int* end;
  int* left;    // global id of the first child (second = left + 1); -1 for a leaf
  int* axis;    // split axis, bucket (valid when left >= 0)
  int* bucket;
  int* size;    // nodes in the subtree
  int* pre;     // pre-order index
};

__device__ inline float axis_of(const float* v, int a) { return v[a]; }

__device__ inline int bucket_of(float c, float lo, float hi) {
  int no = (int)((c - lo) / (hi - lo) * kBuckets);
  return no == kBuckets ? no - 1 : no;
}

__global__ void k_prepare(const float* blo, const float* bhi, float* centre, int* order, int* seg, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int a = 0; a < 3; ++a) centre[3 * i + a] = (blo[3 * i + a] + bhi[3 * i + a]) * 0.5f;  // BBox::centre
  order[i] = i;
  seg[i] = 0;
}

// leaf by size or extent (bvh.cpp build: n <= 4 || an extent < 1e-4); others get empty bins
__global__ void k_classify(NodeArrays N, int first, int count, uint32_t* bins, int* split_flag) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  int node = first + k;
  int n = N.end[node] - N.begin[node];
  float dx = N.hi[3 * node] - N.lo[3 * node], dy = N.hi[3 * node + 1] - N.lo[3 * node + 1],
        dz = N.hi[3 * node + 2] - N.lo[3 * node + 2];
  bool leaf = n <= 4 || dx < 0.0001f || dy < 0.0001f || dz < 0.0001f;
  N.left[node] = leaf ? -1 : -2;  // -2: candidate for a split
  split_flag[k] = 0;
  if (!leaf) {
    uint32_t* b = bins + (size_t)k * kBinInts;
    for (int j = 0; j < 3 * kBuckets; ++j) {
      b[7 * j] = 0u;
      b[7 * j + 1] = b[7 * j + 2] = b[7 * j + 3] = 0xffffffffu;  // lo = "+inf" under atomicMin
      b[7 * j + 4] = b[7 * j + 5] = b[7 * j + 6] = 0u;           // hi = "-inf" under atomicMax
    }
  }
}

__device__ inline void bin_add(uint32_t* b, int a, int bucket, const float* lo, const float* hi) {
  uint32_t* e = b + 7 * (a * kBuckets + bucket);
  atomicAdd(e, 1u);
  atomicMin(e + 1, f2key(lo[0])); atomicMin(e + 2, f2key(lo[1])); atomicMin(e + 3, f2key(lo[2]));
  atomicMax(e + 4, f2key(hi[0])); atomicMax(e + 5, f2key(hi[1])); atomicMax(e + 6, f2key(hi[2]));
}

__global__ __launch_bounds__(kBlock) void k_bin(NodeArrays N, int first, const int* order, const int* seg, const float* blo,
                                                const float* bhi, const float* centre, uint32_t* bins, int n) {
  __shared__ uint32_t s_bins[kBinInts];
  __shared__ int s_node;
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (threadIdx.x == 0) s_node = seg[blockIdx.x * kBlock < n ? blockIdx.x * kBlock : n - 1];
  for (int j = threadIdx.x; j < kBinInts; j += kBlock) {
    int f = j % 7;
    s_bins[j] = f == 0 ? 0u : (f <= 3 ? 0xffffffffu : 0u);
  }
  __syncthreads();
  const int node0 = s_node;
  if (i < n) {
    int node = seg[i];
    if (node >= first && N.left[node] == -2) {
      int p = order[i];
      const float* lo = blo + 3 * p;
      const float* hi = bhi + 3 * p;
      uint32_t* b = node == node0 ? s_bins : bins + (size_t)(node - first) * kBinInts;
      for (int a = 0; a < 3; ++a) {
        int no = bucket_of(centre[3 * p + a], N.lo[3 * node + a], N.hi[3 * node + a]);
        bin_add(b, a, no, lo, hi);
      }
    }
  }
  __syncthreads();
  if (node0 >= first && N.left[node0] == -2) {
    uint32_t* g = bins + (size_t)(node0 - first) * kBinInts;
    for (int j = threadIdx.x; j < kBinInts; j += kBlock) {
      int f = j % 7;
      uint32_t v = s_bins[j];
      if (f == 0) { if (v) atomicAdd(g + j, v); }
      else if (f <= 3) { if (v != 0xffffffffu) atomicMin(g + j, v); }
      else { if (v != 0u) atomicMax(g + j, v); }
    }
  }
}

struct Box3 {
  float lo[3], hi[3];
  __device__ void clear() { for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; } }
  __device__ void expand(const Box3& b) {
    for (int a = 0; a < 3; ++a) { lo[a] = fps::fmin_(b.lo[a], lo[a]); hi[a] = fps::fmax_(b.hi[a], hi[a]); }
  }
  __device__ float area() const {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return 2.f * (dx * dy + dy * dz + dz * dx);
  }
};

// The candidate-plane search of the host build, one thread per node.
__global__ void k_decide(NodeArrays N, int first, int count, const uint32_t* bins, int* split_flag, float* child_boxes) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  int node = first + k;
  if (N.left[node] != -2) return;
  const uint32_t* b = bins + (size_t)k * kBinInts;
  int n = N.end[node] - N.begin[node];
  Box3 box;
  for (int a = 0; a < 3; ++a) { box.lo[a] = N.lo[3 * node + a]; box.hi[a] = N.hi[3 * node + a]; }
  int best_axis = -1, best_bucket = 0;
  float best_cost = (float)n * box.area();
  for (int a = 0; a < 3; ++a) {
    for (int j = 1; j < kBuckets; ++j) {
      Box3 b0, b1;
      b0.clear(); b1.clear();
      int c0 = 0, c1 = 0;
      for (int q = 0; q < kBuckets; ++q) {
        const uint32_t* e = b + 7 * (a * kBuckets + q);
        Box3 bb;
        for (int x = 0; x < 3; ++x) { bb.lo[x] = key2f(e[1 + x]); bb.hi[x] = key2f(e[4 + x]); }
        if (e[0] == 0u) bb.clear();
        if (q < j) { b0.expand(bb); c0 += (int)e[0]; } else { b1.expand(bb); c1 += (int)e[0]; }
      }
      float sa = (c0 == 0) ? 0.f : b0.area() * (float)c0;
      float sb = (c1 == 0) ? 0.f : b1.area() * (float)c1;
      float cost = sa + sb;
      if (cost < best_cost) { best_cost = cost; best_axis = a; best_bucket = j; }
    }
  }
  if (best_axis == -1) { N.left[node] = -1; return; }
  N.axis[node] = best_axis;
  N.bucket[node] = best_bucket;
  N.left[node] = -3;  // split decided; child ids follow from the scan of split_flag
  split_flag[k] = 1;
  Box3 l, r;
  l.clear(); r.clear();
  for (int q = 0; q < kBuckets; ++q) {
    const uint32_t* e = b + 7 * (best_axis * kBuckets + q);
    if (e[0] == 0u) continue;
    Box3 bb;
    for (int x = 0; x < 3; ++x) { bb.lo[x] = key2f(e[1 + x]); bb.hi[x] = key2f(e[4 + x]); }
    if (q < best_bucket) l.expand(bb); else r.expand(bb);
  }
  float* cb = child_boxes + (size_t)k * 12;
  for (int x = 0; x < 3; ++x) { cb[x] = l.lo[x]; cb[3 + x] = l.hi[x]; cb[6 + x] = r.lo[x]; cb[9 + x] = r.hi[x]; }
}

__global__ void k_flags(NodeArrays N, int first, const int* order, const int* seg, const float* centre, int* flag, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int node = seg[i];
  int f = 0;
  if (node >= first && N.left[node] == -3) {
    int a = N.axis[node];
    f = bucket_of(centre[3 * order[i] + a], N.lo[3 * node + a], N.hi[3 * node + a]) < N.bucket[node] ? 1 : 0;
  }
  flag[i] = f;
}

// children of the split nodes of this level; `rank` = exclusive scan of split_flag
__global__ void k_children(NodeArrays N, int first, int count, const int* split_flag, const int* rank, const int* flag,
                           const int* scan, const float* child_boxes, int next_first) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count || !split_flag[k]) return;
  int node = first + k;
  int b = N.begin[node], e = N.end[node];
  int nl = scan[e - 1] + flag[e - 1] - scan[b];
  int l = next_first + 2 * rank[k], r = l + 1;
  N.left[node] = l;
  const float* cb = child_boxes + (size_t)k * 12;
  for (int x = 0; x < 3; ++x) {
    N.lo[3 * l + x] = cb[x]; N.hi[3 * l + x] = cb[3 + x];
    N.lo[3 * r + x] = cb[6 + x]; N.hi[3 * r + x] = cb[9 + x];
  }
  N.begin[l] = b; N.end[l] = b + nl;
  N.begin[r] = b + nl; N.end[r] = e;
}

// stable partition of every split segment
__global__ void k_scatter(NodeArrays N, int first, const int* order, const int* seg, const int* flag, const int* scan,
                          int* order_out, int* seg_out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int node = seg[i];
  int l = node >= first ? N.left[node] : -1;
  if (l < 0) { order_out[i] = order[i]; seg_out[i] = node; return; }
  int b = N.begin[node], e = N.end[node];
  int before = scan[i] - scan[b];
  int nl = scan[e - 1] + flag[e - 1] - scan[b];
  int pos = flag[i] ? b + before : b + nl + (i - b - before);
  order_out[pos] = order[i];
  seg_out[pos] = flag[i] ? l : l + 1;
}

__global__ void k_sizes(NodeArrays N, int first, int count) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  int node = first + k;
  int l = N.left[node];
  N.size[node] = l < 0 ? 1 : 1 + N.size[l] + N.size[l + 1];
}
__global__ void k_preorder(NodeArrays N, int first, int count) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  int node = first + k;
  int l = N.left[node];
  if (l < 0) return;
  N.pre[l] = N.pre[node] + 1;
  N.pre[l + 1] = N.pre[node] + 1 + N.size[l];
}
__global__ void k_emit(NodeArrays N, int total, Node* out) {
  int node = blockIdx.x * blockDim.x + threadIdx.x;
  if (node >= total) return;
  Node o;
  o.lo = f3{N.lo[3 * node], N.lo[3 * node + 1], N.lo[3 * node + 2]};
  o.hi = f3{N.hi[3 * node], N.hi[3 * node + 1], N.hi[3 * node + 2]};
  int l = N.left[node];
  bool leaf = l < 0;
  bool empty = N.end[node] <= N.begin[node];
  o.second = leaf ? -1 : N.pre[l + 1];
  o.leaf = leaf ? 1 : 0;
  o.start = (leaf && !empty) ? N.begin[node] : -1;
  o.end = (leaf && !empty) ? N.end[node] - 1 : -1;
  out[N.pre[node]] = o;
}

}  // namespace gpubvh

// Builds the tree over `boxes` (one per primitive, in push order) on HIP device
// `device`.  `nodes` receives the pre-order nodes, `order` the primitive indices
// leaf by leaf, exactly as the host build of scene/bvh.cpp delivers them.
bool build_bvh_gpu(int device, const std::vector<BBox>& boxes, const BBox& root_box, std::vector<Node>& nodes,
                   std::vector<int>& order, std::string& err) {
  using namespace gpubvh;
  const int n = (int)boxes.size();
  nodes.clear();
  order.clear();
  if (n == 0) return true;
  BVHCHECK(hipSetDevice(device));
  const bool timing = getenv("CASTER_TIMING") != nullptr;
  const auto t_start = std::chrono::steady_clock::now();
  auto since = [&](std::chrono::steady_clock::time_point t) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count();
  };
  std::vector<void*> allocs;
  auto cleanup = [&]() { for (void* p : allocs) (void)hipFree(p); };
  auto alloc = [&](auto** out, size_t count) {
    using T = std::remove_pointer_t<std::remove_pointer_t<decltype(out)>>;
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T));
    if (e != hipSuccess) return e;
    allocs.push_back(p);
    *out = (T*)p;
    return hipSuccess;
  };
#define BVHTRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); cleanup(); return false; } } while (0)
  const size_t cap = 2 * (size_t)n;           // a binary tree over n non-empty leaves has < 2n nodes
  const size_t level_cap = (size_t)n;         // nodes of one level own disjoint, non-empty segments
  float *blo, *bhi, *centre, *child_boxes;
  int *order_a, *order_b, *seg_a, *seg_b, *flag, *scan, *split_flag, *rank;
  uint32_t* bins;
  NodeArrays N{};
  BVHTRY(alloc(&blo, 3 * (size_t)n)); BVHTRY(alloc(&bhi, 3 * (size_t)n)); BVHTRY(alloc(&centre, 3 * (size_t)n));
  BVHTRY(alloc(&order_a, n)); BVHTRY(alloc(&order_b, n)); BVHTRY(alloc(&seg_a, n)); BVHTRY(alloc(&seg_b, n));
  BVHTRY(alloc(&flag, n)); BVHTRY(alloc(&scan, n));
  BVHTRY(alloc(&split_flag, level_cap)); BVHTRY(alloc(&rank, level_cap));
  BVHTRY(alloc(&child_boxes, 12 * level_cap));
  BVHTRY(alloc(&N.lo, 3 * cap)); BVHTRY(alloc(&N.hi, 3 * cap)); BVHTRY(alloc(&N.begin, cap)); BVHTRY(alloc(&N.end, cap));
  BVHTRY(alloc(&N.left, cap)); BVHTRY(alloc(&N.axis, cap)); BVHTRY(alloc(&N.bucket, cap)); BVHTRY(alloc(&N.size, cap));
  BVHTRY(alloc(&N.pre, cap));
  BVHTRY(alloc(&bins, level_cap * kBinInts));
  {
    std::vector<float> hlo(3 * (size_t)n), hhi(3 * (size_t)n);
    for (int i = 0; i < n; ++i) {
      hlo[3 * i] = boxes[i].lo.x; hlo[3 * i + 1] = boxes[i].lo.y; hlo[3 * i + 2] = boxes[i].lo.z;
      hhi[3 * i] = boxes[i].hi.x; hhi[3 * i + 1] = boxes[i].hi.y; hhi[3 * i + 2] = boxes[i].hi.z;
    }
    BVHTRY(hipMemcpy(blo, hlo.data(), hlo.size() * 4, hipMemcpyHostToDevice));
    BVHTRY(hipMemcpy(bhi, hhi.data(), hhi.size() * 4, hipMemcpyHostToDevice));
  }
  size_t tmp_bytes = 0, tmp2 = 0;
  BVHTRY(rocprim::exclusive_scan(nullptr, tmp_bytes, flag, scan, 0, (size_t)n, rocprim::plus<int>()));
  BVHTRY(rocprim::exclusive_scan(nullptr, tmp2, split_flag, rank, 0, level_cap, rocprim::plus<int>()));
  tmp_bytes = std::max(tmp_bytes, tmp2);
  char* tmp;
  BVHTRY(alloc(&tmp, tmp_bytes));

  const int nb = (n + kBlock - 1) / kBlock;
  const auto t_levels = std::chrono::steady_clock::now();
  const double s_setup = since(t_start);
  hipLaunchKernelGGL(k_prepare, dim3(nb), dim3(kBlock), 0, 0, blo, bhi, centre, order_a, seg_a, n);
  {  // the root
    float lo[3] = {root_box.lo.x, root_box.lo.y, root_box.lo.z}, hi[3] = {root_box.hi.x, root_box.hi.y, root_box.hi.z};
    int zero = 0;
    BVHTRY(hipMemcpy(N.lo, lo, 12, hipMemcpyHostToDevice));
    BVHTRY(hipMemcpy(N.hi, hi, 12, hipMemcpyHostToDevice));
    BVHTRY(hipMemcpy(N.begin, &zero, 4, hipMemcpyHostToDevice));
    BVHTRY(hipMemcpy(N.end, &n, 4, hipMemcpyHostToDevice));
    BVHTRY(hipMemcpy(N.pre, &zero, 4, hipMemcpyHostToDevice));
  }
  std::vector<std::pair<int, int>> levels;  // (first node, count)
  int first = 0, count = 1;
  int* ord_in = order_a; int* ord_out = order_b; int* seg_in = seg_a; int* seg_out = seg_b;
  while (count > 0) {
    if (levels.size() > 4096) { err = "build_bvh_gpu: tree deeper than 4096 levels"; cleanup(); return false; }
    levels.push_back({first, count});
    const int cb = (count + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_classify, dim3(cb), dim3(kBlock), 0, 0, N, first, count, bins, split_flag);
    hipLaunchKernelGGL(k_bin, dim3(nb), dim3(kBlock), 0, 0, N, first, ord_in, seg_in, blo, bhi, centre, bins, n);
    hipLaunchKernelGGL(k_decide, dim3(cb), dim3(kBlock), 0, 0, N, first, count, bins, split_flag, child_boxes);
    BVHTRY(rocprim::exclusive_scan(tmp, tmp_bytes, split_flag, rank, 0, (size_t)count, rocprim::plus<int>()));
    int last_rank = 0, last_flag = 0;
    BVHTRY(hipMemcpy(&last_rank, rank + count - 1, 4, hipMemcpyDeviceToHost));
    BVHTRY(hipMemcpy(&last_flag, split_flag + count - 1, 4, hipMemcpyDeviceToHost));
    const int splits = last_rank + last_flag;
    if (splits == 0) break;
    const int next_first = first + count;
    if ((size_t)next_first + 2 * (size_t)splits > cap) { err = "build_bvh_gpu: node capacity exceeded"; cleanup(); return false; }
    hipLaunchKernelGGL(k_flags, dim3(nb), dim3(kBlock), 0, 0, N, first, ord_in, seg_in, centre, flag, n);
    BVHTRY(rocprim::exclusive_scan(tmp, tmp_bytes, flag, scan, 0, (size_t)n, rocprim::plus<int>()));
    hipLaunchKernelGGL(k_children, dim3(cb), dim3(kBlock), 0, 0, N, first, count, split_flag, rank, flag, scan, child_boxes,
                       next_first);
    hipLaunchKernelGGL(k_scatter, dim3(nb), dim3(kBlock), 0, 0, N, first, ord_in, seg_in, flag, scan, ord_out, seg_out, n);
    std::swap(ord_in, ord_out);
    std::swap(seg_in, seg_out);
    first = next_first;
    count = 2 * splits;
  }
  const int total = levels.back().first + levels.back().second;
  for (size_t l = levels.size(); l-- > 0;)
    hipLaunchKernelGGL(k_sizes, dim3((levels[l].second + kBlock - 1) / kBlock), dim3(kBlock), 0, 0, N, levels[l].first,
                       levels[l].second);
  for (size_t l = 0; l < levels.size(); ++l)
    hipLaunchKernelGGL(k_preorder, dim3((levels[l].second + kBlock - 1) / kBlock), dim3(kBlock), 0, 0, N, levels[l].first,
                       levels[l].second);
  Node* out;
  BVHTRY(alloc(&out, total));
  hipLaunchKernelGGL(k_emit, dim3((total + kBlock - 1) / kBlock), dim3(kBlock), 0, 0, N, total, out);
  BVHTRY(hipGetLastError());
  BVHTRY(hipDeviceSynchronize());
  const double s_levels = since(t_levels);
  const auto t_down = std::chrono::steady_clock::now();
  nodes.resize((size_t)total);
  order.resize((size_t)n);
  BVHTRY(hipMemcpy(nodes.data(), out, (size_t)total * sizeof(Node), hipMemcpyDeviceToHost));
  BVHTRY(hipMemcpy(order.data(), ord_in, (size_t)n * 4, hipMemcpyDeviceToHost));
  cleanup();
  if (timing)
    fprintf(stderr, "caster: GPU BVH build: %zu levels, %d nodes; allocate + upload %.4f s, kernels %.4f s (%.1f M primitives/s), download + free %.4f s\n",
            levels.size(), total, s_setup, s_levels, n / s_levels / 1e6, since(t_down));
  return true;
#undef BVHTRY
}

}  // namespace caster
