// Part of the gfx950 wavefront renderer (included by ../kernels.hip, in this order: common.h, traverse.h, trace.h, shading.h, bsdf.h, path.h, volpath.h, lighttrace.h, bdpt.h, volpath_queues.h, film.h).
// Vectors, the per-pixel random stream, sampling warps, primary rays.
#pragma once

namespace caster {

constexpr int kWave = 64;
#ifndef CASTER_TWO_RAYS
#define CASTER_TWO_RAYS 0   // the traversal kernel holds two rays per lane (see k_trace)
#endif
#ifndef CASTER_LDS_DEPTH
#define CASTER_LDS_DEPTH (CASTER_TWO_RAYS ? 7 : 10)
#endif
#ifndef CASTER_REFILL
#define CASTER_REFILL (CASTER_TWO_RAYS ? 40 : 24)
#endif
#ifndef CASTER_CHUNK
#define CASTER_CHUNK 64
#endif
#ifndef CASTER_SHARDS
#define CASTER_SHARDS 8
#endif
#ifndef CASTER_VOTE_NUM  // do node work when n_inner * NUM >= n_leaf * DEN
#define CASTER_VOTE_NUM 1
#define CASTER_VOTE_DEN 1
#endif
#ifndef CASTER_TRACE_BLOCK_WAVES
#define CASTER_TRACE_BLOCK_WAVES 1
#endif
#ifndef CASTER_TOP_NODES
#define CASTER_TOP_NODES 21
#endif
constexpr int kTraceBlockWaves = CASTER_TRACE_BLOCK_WAVES;  // wavefronts per traversal workgroup (they share the LDS copy of the top of the tree)
constexpr int kTopNodes = CASTER_TOP_NODES;                 // 4-wide nodes, in breadth-first order from the root, kept in LDS
constexpr int kLdsDepth = CASTER_LDS_DEPTH;  // traversal-stack entries per lane kept in LDS
constexpr int kSpillDepth = 48;              // further entries in a per-lane column in HBM
constexpr uint32_t kChunk = CASTER_CHUNK;        // rays a wavefront claims from a queue per atomic
constexpr uint32_t kWorkShards = CASTER_SHARDS;  // work counters per queue
constexpr uint32_t kWorkPad = 64;                 // counters sit 256 bytes apart: one memory channel each
constexpr int kRefillIdle = CASTER_REFILL;   // refill a wavefront once this many lanes are idle


// ------------------------------------------------------------------ vectors
struct V3 { float x, y, z; };
struct V2 { float x, y; };
__device__ inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ inline V3 v3(float4 f) { return V3{f.x, f.y, f.z}; }
__device__ inline V3 v3(const float* p) { return V3{p[0], p[1], p[2]}; }
__device__ inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
__device__ inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
__device__ inline V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
__device__ inline V3 operator/(V3 a, V3 b) { return V3{a.x / b.x, a.y / b.y, a.z / b.z}; }
__device__ inline V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
__device__ inline V3 operator+(V3 a, float s) { return V3{a.x + s, a.y + s, a.z + s}; }
__device__ inline V3 operator-(V3 a, float s) { return V3{a.x - s, a.y - s, a.z - s}; }
__device__ inline V3 operator-(float s, V3 a) { return V3{s - a.x, s - a.y, s - a.z}; }
// (fused as nvcc's default contraction fuses cutil_math.h:1136, 1308: see fpspec.h)
__device__ inline float dot(V3 a, V3 b) { return fps::dot3_(a.x, a.y, a.z, b.x, b.y, b.z); }
__device__ inline V3 cross(V3 a, V3 b) {
  return V3{fps::diffprod_(a.y, b.z, a.z, b.y), fps::diffprod_(a.z, b.x, a.x, b.z), fps::diffprod_(a.x, b.y, a.y, b.x)};
}
__device__ inline float length(V3 v) { return sqrtf(dot(v, v)); }
__device__ inline V3 normalize(V3 v) { return v * fps::rsqrt_(dot(v, v)); }
__device__ inline V3 vabs(V3 v) { return V3{fabsf(v.x), fabsf(v.y), fabsf(v.z)}; }
__device__ inline float maxcomp(V3 a) { return fps::fmax_(fps::fmax_(a.x, a.y), a.z); }
// The lanes of the wavefront for which `p` holds, as the compare's own result mask (HIP's __ballot takes an int:
// the flag is first made a 0/1 number and then compared with zero again, two vector instructions per vote).
__device__ inline unsigned long long lane_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ inline bool any_lane(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
// maxcomp(vabs(a)) as one v_max3_f32 with |.| operands: among non-negative numbers the native maximum picks the value
// fps::fmax_ picks (it drops a NaN operand as fmax_ does, and no zero here carries a sign).
__device__ inline float maxabs3(V3 a) { return __builtin_fmaxf(__builtin_fmaxf(fabsf(a.x), fabsf(a.y)), fabsf(a.z)); }
__device__ inline bool is_black(V3 c) { return c.x == 0 && c.y == 0 && c.z == 0; }
__device__ inline bool isinf_(float v) { return v == INFINITY || v == -INFINITY; }
__device__ inline bool has_nan(V3 c) { return c.x != c.x || c.y != c.y || c.z != c.z; }
__device__ inline bool has_inf(V3 c) { return isinf_(c.x) || isinf_(c.y) || isinf_(c.z); }
__device__ inline float4 f4(V3 v, float w) { return make_float4(v.x, v.y, v.z, w); }
__device__ inline float4 f4(V3 v, uint32_t w) { return make_float4(v.x, v.y, v.z, __uint_as_float(w)); }

// common.h:17-22 of the reference
constexpr float PI = 3.14159265358f;
constexpr float TWOPI = 6.28318530716f;
constexpr float ONE_OVER_PI = 0.3183098861847f;
constexpr float ONE_OVER_TWO_PI = 0.1591549430923f;
constexpr float ONE_OVER_FOUR_PI = 0.0795774715461f;

// ---------------------------------------------------------------------- RNG
// thrust::default_random_engine (minstd_rand: x <- 48271 x mod 2^31-1) seeded
// with WangHash(pixel)+WangHash(iter) (pathtracer.cu:830, common.h:200-209).
__device__ inline uint32_t wang_hash(uint32_t seed) {
  seed = (seed ^ 61u) ^ (seed >> 16);
  seed = seed + (seed << 3);
  seed = seed ^ (seed >> 4);
  seed = seed * 0x27d4eb2du;
  seed = seed ^ (seed >> 15);
  return seed;
}
__device__ inline uint32_t rng_seed(uint32_t pixel, uint32_t iter) {
  uint32_t s = wang_hash(pixel) + wang_hash(iter);
  uint32_t x = s % 2147483647u;
  return x == 0 ? 1u : x;
}
__device__ inline float rng_uniform(uint32_t& x) {
  // 2^31-1 is a Mersenne prime: fold the 47-bit product instead of dividing.
  uint64_t p = (uint64_t)x * 48271u;
  uint32_t r = (uint32_t)(p & 0x7fffffffu) + (uint32_t)(p >> 31);
  if (r >= 2147483647u) r -= 2147483647u;
  x = r;
  float f = (float)(r - 1u);
  f /= 2147483648.0f;  // 1.0f + float(2147483645)
  return f;
}

// ---------------------------------------------------------- sampling warps
__device__ inline V3 cosine_hemisphere(float u1, float u2, float& pdf) {  // Sampling.h:31-43
  float sintheta = sqrtf(u1);
  float costheta = sqrtf(1.f - u1);
  float phi = TWOPI * u2;
  float cosphi = fps::cos_(phi);
  float sinphi = fps::sin_(phi);
  pdf = fabsf(costheta) * ONE_OVER_PI;
  return v3(sintheta * cosphi, sintheta * sinphi, costheta);
}
__device__ inline V3 uniform_sphere(float u1, float u2, float& pdf) {  // Sampling.h:4-15
  float costheta = 1.f - 2.f * u1;
  float sintheta = sqrtf(1.f - costheta * costheta);
  float phi = TWOPI * u2;
  float cosphi = fps::cos_(phi);
  float sinphi = fps::sin_(phi);
  pdf = ONE_OVER_FOUR_PI;
  return v3(sintheta * cosphi, sintheta * sinphi, costheta);
}
__device__ inline V3 to_world(V3 d, V3 u, V3 v, V3 w) { return d.x * u + d.y * v + d.z * w; }
__device__ inline void make_coordinate(V3 n, V3& u, V3& w) {  // common.h:116-129
  if (fabsf(n.x) > fabsf(n.y)) {
    float invLen = 1.0f / sqrtf(n.x * n.x + n.z * n.z);
    w = v3(n.z * invLen, 0.0f, -n.x * invLen);
  } else {
    float invLen = 1.0f / sqrtf(n.y * n.y + n.z * n.z);
    w = v3(0.0f, n.z * invLen, -n.y * invLen);
  }
  u = cross(w, n);
}

// ------------------------------------------------------------- primary rays
// Camera::GeneratePrimaryRay (Camera.h:45-85)
__device__ inline void primary_ray(const DevCamera& c, float x, float y, float a1, float a2, V3& orig, V3& dir) {
  V3 cu = v3(c.u), cv = v3(c.v), cw = v3(c.w);
  orig = v3(c.pos);
  if (c.environment) {
    float theta = PI * (1.f - y / c.res_y);
    float phi = TWOPI * (1.f - x / c.res_x);
    V3 l = v3(fps::sin_(theta) * fps::cos_(phi), fps::cos_(theta), fps::sin_(theta) * fps::sin_(phi));
    dir = l.x * cu + l.y * cv - l.z * cw;
    return;
  }
  float xx = x * c.p2s_x - c.half_w;
  float yy = y * c.p2s_y - c.half_h;
  if (c.apertureRadius > 0.00001f) {
    // UniformSampleDisk (Sampling.h:59-67)
    float r = sqrtf(a1);
    float phi = TWOPI * a2;
    float ax = (r * fps::cos_(phi)) * c.apertureRadius;
    float ay = (r * fps::sin_(phi)) * c.apertureRadius;
    float focal_x = c.ratio * xx;
    float focal_y = c.ratio * yy;
    V3 aperture = v3(ax, ay, 0);
    V3 focal = v3(focal_x, focal_y, -c.focalDistance);
    V3 l = focal - aperture;
    dir = l.x * cu + l.y * cv + l.z * cw;
    orig = orig + (aperture.x * cu + aperture.y * cv);
  } else {
    dir = xx * cu + yy * cv + -c.distance * cw;
  }
  dir = normalize(dir);
}

// Path slot <-> pixel: slots of one iteration are laid out tile by tile so a
// wavefront's 64 primary rays cover an 8x8 pixel block.
__device__ inline bool row_live(const DevScene& S, uint32_t py) {
  if ((int)py < S.row_y0 || (int)py >= S.row_y1) return false;
  return (py - (uint32_t)S.row_y0) % (uint32_t)S.row_stride < (uint32_t)S.row_rows;
}
// The pixel of slot `local`; false when the slot lies outside the image or on a scanline this renderer does not trace.
__device__ inline bool slot_pixel(const DevScene& S, uint32_t local, uint32_t& px, uint32_t& py) {
  uint32_t tile = local >> 6, in = local & 63u;
  uint32_t tx = tile % (uint32_t)S.tiles_x, e = tile / (uint32_t)S.tiles_x;
  uint32_t ty = (uint32_t)S.tile_row0 + (e / (uint32_t)S.tile_rows) * (uint32_t)S.tile_stride + e % (uint32_t)S.tile_rows;
  px = tx * 8 + (in & 7u);
  py = ty * 8 + (in >> 3);
  return px < (uint32_t)S.width && py < (uint32_t)S.height && row_live(S, py);
}
// The slot of a pixel on a traced scanline.
__device__ inline uint32_t pixel_slot(const DevScene& S, uint32_t px, uint32_t py) {
  uint32_t d = (py >> 3) - (uint32_t)S.tile_row0;
  uint32_t e = (d / (uint32_t)S.tile_stride) * (uint32_t)S.tile_rows + d % (uint32_t)S.tile_stride;
  uint32_t tile = e * (uint32_t)S.tiles_x + (px >> 3);
  return (tile << 6) | ((py & 7u) << 3) | (px & 7u);
}

}  // namespace caster
