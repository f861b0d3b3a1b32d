// Device-side scene and path-state layout of the gfx950 wavefront renderer.
//
// The reference uploads its host structs verbatim (40-byte nodes, 176-byte
// tagged-union primitives, 192-byte lights: CUDA-PathTracer/tracer/
// pathtracer.cu:2554-2681) and binds them to __device__ globals with a <<<1,1>>>
// kernel (pathtracer.cu:2519-2552).  Here everything a kernel needs is one
// by-value parameter block of pointers into 16-byte-aligned float4 arrays laid
// out for what each stage reads:
//
//   nodes   8 x float4 per 4-wide inner node: lo.x[4] lo.y[4] lo.z[4] hi.x[4]
//           hi.y[4] hi.z[4] child-ref[4] (pad) -- one 128-byte fetch decides
//           four box tests; the reference fetches a 40-byte node, tests its
//           own box, then fetches each of two children
//   isect   3 x float4 per primitive: only what the ray/primitive test reads
//   shade   8 x float4 per primitive: normals, uvs, tangents, indices; read
//           once per accepted hit by the shade stage
//   lights  6 x float4 per emissive triangle
//   mats    6 x float4 per material
//   media   5 x float4 per medium: grid size | g, sigmaS | type, sigmaT |
//           1/max density, p0 | offset into `density`, p1 | iteration cap
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace caster {

constexpr int kNodeF4 = 8;
constexpr int kIsectF4 = 3;
constexpr int kShadeF4 = 8;
constexpr int kLightF4 = 6;
constexpr int kMatF4 = 6;
constexpr int kMediumF4 = 5;

constexpr int kRefEmpty = (int)0x80000000;  // child reference of an empty subtree

// bits kept in the .w lane of the third isect float4
constexpr uint32_t kPrimTypeMask = 3u;      // GT_TRIANGLE / GT_LINES / GT_SPHERE
constexpr uint32_t kPrimLastInLeaf = 4u;    // leaf ranges end at the primitive carrying this bit

struct DevCamera {
  float pos[3], u[3], v[3], w[3];
  float res_x, res_y, distance, apertureRadius, focalDistance;
  float half_w, half_h, p2s_x, p2s_y, ratio, area;
  int environment, medium, filmic;
};

struct DevInfinite {
  int valid, w, h;
  float radius;
  float u[3], v[3], w_[3];
  const float* data;  // 3 floats per texel, row-major
};

struct DevScene {
  const float4* nodes;
  const float4* isect;
  const float4* shade;
  const float4* lights;
  const float4* mats;
  const float4* media;
  const float* density;  // density grids of the heterogeneous media, back to back
  int has_hetero;        // some medium is a density grid (its transmittance draws random numbers)
  const float* light_cdf;
  const uint32_t* tex_pool;   // packed rgba8 texels of every texture
  const int* tex_info;        // per texture: offset, width, height
  int n_prims, n_nodes, n_lights, n_cdf;
  DevCamera cam;
  DevInfinite inf;
  float epsilon;
  float g1, g2, g3;     // pbrt gamma(1..3), as the reference computes them
  float box_scale;      // 1 + 2*gamma(3)
  float far_t;          // no hit is looked for farther along a ray than this (Scene::far_t)
  float emit_lo[3], emit_hi[3];  // box around all emissive triangles, padded far beyond rounding error
  int integrator, maxDepth;
  float maxDist;
  int width, height;
  // The scanlines this renderer traces (sharding across GPUs): y0 <= y < y1 and (y - y0) % stride < rows, i.e.
  // slabs of `rows` scanlines every `stride`; the whole image is (0, height, 1, 1).
  int row_y0, row_y1, row_rows, row_stride;
  // Path slots of one iteration are enumerated tile by tile (8x8 pixels: one tile per wavefront at generation)
  // over the tile rows tile_row0 + (e / tile_rows) * tile_stride + e % tile_rows, e = 0 .. tiles_y - 1: every
  // tile row of the image (0, 1, 1), or, where the scanline slabs are made of whole tile rows, only theirs,
  // so that a renderer with an eighth of the image spends its path slots on an eighth of the image.
  int tile_row0, tile_rows, tile_stride;
  int tiles_x, tiles_y;   // tiles per tile row; tile rows enumerated
  int slots_per_iter;     // tiles_x * tiles_y * 64
};

// Per-path state, structure of float4 arrays indexed by path slot.
//   beta_rng  beta.xyz, minstd_rand state (uint bits)
//   li_flags  Li.xyz, flags (uint bits, see kF*)
//   pend_x    light-sampled direct term of the last shaded vertex (zeroed by
//             the shadow stage when occluded), .w = |dot(wi, n)| of the BSDF-sampled probe
//   pend_y    BSDF-sampled direct term: (fr.xyz, pdf) until the probe stage
//             replaces it with the weighted contribution
//   beta_prev throughput the two pending terms are to be multiplied with
struct PathState {
  float4* beta_rng;
  float4* li_flags;
  float4* pend_x;
  float4* pend_y;
  float4* beta_prev;
  // volumetric integrator on queues only (null otherwise): operands of the light-sampled term,
  // which can only be evaluated once the transmittance walk of the shadow ray is over
  float4* pend_z;  // radiance | lightPdf * choicePdf
  float4* pend_w;  // transmittance so far | medium of the shadow ray's current segment
  float4* pend_v;  // |cos| of the probe direction, |cos| of the light direction, medium of the vertex's ray
};

constexpr uint32_t kFBounceMask = 0xffffu;       // bounce counter
constexpr uint32_t kFSpecular = 1u << 16;
constexpr uint32_t kFPending = 1u << 17;
constexpr uint32_t kFPendingY = 1u << 18;        // pend_y holds a BSDF-sampled term (else it counts as zero)
constexpr uint32_t kFPendingM = 1u << 19;        // vpt: the pending term is that of a scattering event in a medium
constexpr uint32_t kFPendingX = 1u << 28;        // vpt: the pending term has a light-sampled part
constexpr uint32_t kFMediumShift = 20;           // bits 20..27: medium index + 1
constexpr uint32_t kFMediumMask = 0xffu << kFMediumShift;

// A compacted queue of rays: origin.xyz,tmin | dir.xyz,tmax | path slot, and,
// for extension rays, the hit the extend stage found (t, prim id, b1, b2).
struct RayQueue {
  float4* o;
  float4* d;
  int* slot;
  float4* hit;
  uint32_t* count;
};

struct Stats {
  unsigned long long rays[4];       // extension, shadow, MIS probe, transmittance segment
  unsigned long long samples;
  unsigned long long invalid;
  unsigned long long stack_overflow;
  // k_trace's own work, [closest hit, any hit]: 4-wide node records fetched from memory (the top of
  // the tree is served from LDS) and primitive records tested
  unsigned long long node_fetches[2];
  unsigned long long prim_tests[2];
};

}  // namespace caster
