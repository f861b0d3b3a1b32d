// gfx950 wavefront path tracer: generate / extend / shade / shadow / probe /
// finish kernels over compacted ray queues.
//
// One path per lane, one 64-lane wavefront per workgroup.  A path's state
// lives in SoA float4 arrays in HBM and travels between stages by slot index;
// rays travel in compacted queues written with one atomic per wavefront.  The
// closest-hit and any-hit walks keep a per-lane stack in LDS, lane-strided so
// that a push or pop is one conflict-free ds access.
//
// What each stage computes follows the reference integrators
// (CUDA-PathTracer/tracer/pathtracer.cu:822-1228) exactly, sample by sample:
// the same per-pixel minstd_rand stream, the same draw order, the same
// arithmetic in the same order (built with -ffp-contract=off against
// fpspec.h), so an image is a pure function of (scene, pixel, iteration).
#include <hip/hip_runtime.h>

#include "../fpspec.h"
#include "device_scene.h"

namespace caster {

constexpr int kWave = 64;
#ifndef CASTER_LDS_DEPTH
#define CASTER_LDS_DEPTH 10
#endif
#ifndef CASTER_REFILL
#define CASTER_REFILL 24
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
__device__ inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ inline V3 cross(V3 a, V3 b) {
  return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ inline float length(V3 v) { return sqrtf(dot(v, v)); }
__device__ inline V3 normalize(V3 v) { return v * fps::rsqrt_(dot(v, v)); }
__device__ inline V3 vabs(V3 v) { return V3{fabsf(v.x), fabsf(v.y), fabsf(v.z)}; }
__device__ inline float maxcomp(V3 a) { return fps::fmax_(fps::fmax_(a.x, a.y), a.z); }
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
__device__ inline bool slot_pixel(const DevScene& S, uint32_t local, uint32_t& px, uint32_t& py) {
  uint32_t tile = local >> 6, in = local & 63u;
  uint32_t tx = tile % (uint32_t)S.tiles_x, ty = tile / (uint32_t)S.tiles_x;
  px = tx * 8 + (in & 7u);
  py = ty * 8 + (in >> 3);
  return px < (uint32_t)S.width && py < (uint32_t)S.height;
}
__device__ inline uint32_t pixel_slot(const DevScene& S, uint32_t px, uint32_t py) {
  uint32_t tile = (py >> 3) * (uint32_t)S.tiles_x + (px >> 3);
  return (tile << 6) | ((py & 7u) << 3) | (px & 7u);
}

// --------------------------------------------------------------- traversal
// Slab test of the reference (bbox.h:75-98): tFar widened by 1+2*gamma(3),
// rejected when behind the ray, empty, or beyond the current tmax; tmin is
// not consulted.  Returns tNear through `tn`.
__device__ inline bool box_test(V3 lo, V3 hi, V3 o, V3 inv, float tmax, float scale, float& tn) {
  float txNear = (lo.x - o.x) * inv.x;
  float txFar = (hi.x - o.x) * inv.x;
  float tyNear = (lo.y - o.y) * inv.y;
  float tyFar = (hi.y - o.y) * inv.y;
  float tzNear = (lo.z - o.z) * inv.z;
  float tzFar = (hi.z - o.z) * inv.z;
  // v_min_f32 / v_max_f32 drop a NaN operand exactly as fps::fmin_/fmax_ do;
  // they may differ in the sign of a zero result, which no comparison below sees.
  float tNear = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(txNear, txFar), __builtin_fminf(tyNear, tyFar)),
                                __builtin_fminf(tzNear, tzFar));
  float tFar = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(txNear, txFar), __builtin_fmaxf(tyNear, tyFar)),
                               __builtin_fmaxf(tzNear, tzFar));
  tFar *= scale;
  tn = tNear;
  return !(tFar <= 0.00001f) && !(tNear > tFar) && !(tNear > tmax);
}

struct Hit {
  float t;
  int prim;
  float b1, b2;  // triangle barycentrics; line: s, d2
};

// Ray/primitive tests.  Each returns the candidate parameter in `t`; the
// caller applies the closest-hit rule.  Triangle: Triangle.h:87-118 with the
// edge vectors and coordinate maxima precomputed per primitive (the same
// float operations, done once on the host).  Sphere: sphere.h:24-66.  Line:
// line.h:33-72.
__device__ inline bool prim_test_rec(const DevScene& S, float4 a, float4 b, float4 c, V3 o, V3 d, float maxD, float tmin,
                                     float tmax, float& t, float& ob1, float& ob2, bool& last);
__device__ inline bool prim_test(const DevScene& S, int i, V3 o, V3 d, float maxD, float tmin, float tmax,
                                 float& t, float& ob1, float& ob2, bool& last) {
  const float4* rec = S.isect + (size_t)i * kIsectF4;
  return prim_test_rec(S, rec[0], rec[1], rec[2], o, d, maxD, tmin, tmax, t, ob1, ob2, last);
}
// The same test on a record already in registers.
__device__ inline bool prim_test_rec(const DevScene& S, float4 a, float4 b, float4 c, V3 o, V3 d, float maxD, float tmin,
                                     float tmax, float& t, float& ob1, float& ob2, bool& last) {
  uint32_t bits = __float_as_uint(c.w);
  last = (bits & kPrimLastInLeaf) != 0;
  uint32_t type = bits & kPrimTypeMask;
  if (type == 0u) {
    V3 p1 = v3(a), e1 = v3(b), e2 = v3(c);
    float maxE = a.w, maxP1 = b.w;
    V3 s1 = cross(d, e2);
    float divisor = dot(s1, e1);
    if (fabsf(divisor) < 1e-8f) return false;
    float inv = 1.0f / divisor;
    V3 s = o - p1;
    float b1 = dot(s, s1) * inv;
    if (b1 < 0.0f || b1 > 1.0f) return false;
    V3 s2 = cross(s, e1);
    float b2 = dot(d, s2) * inv;
    if (b2 < 0.0f || b1 + b2 > 1.0f) return false;
    float deltaE = S.g1 * maxE;
    float deltaS1 = 2 * (S.g2 * maxD * maxE + deltaE * maxD);
    float deltaS = S.g1 * maxP1;
    float maxS = maxcomp(vabs(s));
    float maxS1 = maxcomp(vabs(s1));
    float deltaS2 = 2 * (S.g2 * maxS * maxS1 + deltaS * maxS1 + deltaS1 * maxS);
    float maxS2 = maxcomp(vabs(s2));
    float deltaT = 2 * (S.g2 * maxE * maxS2 + deltaS2 * maxE + deltaE * maxS2) * fabsf(inv);
    float tt = dot(e2, s2) * inv;
    if (tt < tmin + deltaT || tt > tmax) return false;
    t = tt; ob1 = b1; ob2 = b2;
    return true;
  }
  if (type == 2u) {
    V3 op = o - v3(a);
    float radius = a.w;
    float B = dot(op, d);
    float C = dot(op, op) - radius * radius;
    float delta = B * B - C;
    if (delta < 0.f) return false;
    float sq = sqrtf(delta);
    float t1 = -B - sq;
    float t2 = -B + sq;
    if (t1 < 0.f && t2 < 0.f) return false;
    if (t1 < 0.f || t2 < 0.f) {
      float tt1 = t1, tt2 = t2;
      t1 = tt1 < 0.f ? tt2 : tt1;
      t2 = tt1 < 0.f ? tt1 : tt2;
    } else if (t1 > t2) {
      float tmp = t2; t2 = t1; t1 = tmp;
    }
    if (t1 > tmax) return false;
    float th;
    if (t1 > tmin) th = t1;
    else if (t2 > 0.f) th = t2;
    else return false;
    if (th > tmax) return false;
    t = th; ob1 = 0.f; ob2 = 0.f;
    return true;
  }
  {
    V3 p0 = v3(a), p1 = v3(b);
    V3 u = d;
    V3 v = p1 - p0;
    V3 w = o - p0;
    float aa = dot(u, u), bb = dot(u, v), cc = dot(v, v), dd = dot(u, w), ee = dot(v, w);
    float det = aa * cc - bb * bb;
    if (det == 0) return false;
    float tt = (bb * ee - cc * dd) / det;
    float s = (aa * ee - bb * dd) / det;
    if (tt < tmin || tt > tmax) return false;
    s = fps::clamp_(s, 0.f, 1.f);
    V3 pr = o + d * tt;
    V3 pl = p0 + (p1 - p0) * s;
    V3 prl = pr - pl;
    float d2 = dot(prl, prl);
    float r = a.w * (1 - s) + b.w * s;
    if (d2 > r * r) return false;
    t = tt; ob1 = s; ob2 = d2;
    return true;
  }
}

// Traversal stack: (child reference, entry distance) pairs.  The first
// kLdsDepth entries of a lane live in LDS, lane-strided so that one wavefront
// push or pop is a single conflict-free 8-byte ds access; deeper entries go to
// a per-lane column in HBM (lane-strided too, so a wavefront's spill is one
// coalesced store).
struct WalkAux {
  uint32_t* work;         // kWorkShards counters: next unclaimed ray of each part of the queue being traced
  uint2* spill;           // kSpillDepth x spill_stride entries
  uint32_t spill_stride;  // lanes in the launch grid
};
struct Stack {
  uint2 (*lds)[kWave];
  uint2* spill;  // this lane's column
  uint32_t stride;
  int sp;
};
#ifndef CASTER_STACK_SPLIT
#define CASTER_STACK_SPLIT 1
#endif
#if CASTER_STACK_SPLIT
// The LDS entries are accessed as one volatile 64-bit word: that keeps each push
// and pop a single ds_write_b64 / ds_read_b64, where the optimiser would
// otherwise merge the LDS and the spill branch into 32-bit flat accesses.
typedef __attribute__((address_space(3))) volatile unsigned long long lds_u64_t;
__device__ inline void lds_put(Stack& s, int sp, uint32_t lane, uint2 e) {
  *(lds_u64_t*)&s.lds[sp][lane] = ((unsigned long long)e.y << 32) | e.x;
}
__device__ inline uint2 lds_get(Stack& s, int sp, uint32_t lane) {
  unsigned long long v = *(lds_u64_t*)&s.lds[sp][lane];
  return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}
#else
__device__ inline void lds_put(Stack& s, int sp, uint32_t lane, uint2 e) { s.lds[sp][lane] = e; }
__device__ inline uint2 lds_get(Stack& s, int sp, uint32_t lane) { return s.lds[sp][lane]; }
#endif
__device__ inline bool stack_push(Stack& s, uint32_t lane, int ref, float tn) {
  uint2 e = make_uint2((uint32_t)ref, __float_as_uint(tn));
  if (s.sp < kLdsDepth) lds_put(s, s.sp, lane, e);
  else if (s.sp < kLdsDepth + kSpillDepth) s.spill[(size_t)(s.sp - kLdsDepth) * s.stride] = e;
  else return false;
  ++s.sp;
  return true;
}
// Next deferred child that can still hold a closer hit: an entry whose entry
// distance exceeds the current tmax is dropped, which is the test the
// reference makes when it reaches that node (pathtracer.cu:231-232,
// bbox.h:95).  kRefEmpty when nothing is left.
__device__ inline int stack_next(Stack& s, uint32_t lane, float tmax) {
  while (s.sp > 0) {
    --s.sp;
    uint2 e;
    if (s.sp < kLdsDepth) e = lds_get(s, s.sp, lane);
    else e = s.spill[(size_t)(s.sp - kLdsDepth) * s.stride];
    if (!(__uint_as_float(e.y) > tmax)) return (int)e.x;
  }
  return kRefEmpty;
}

// One 4-wide node: four slab tests, the children that pass sorted nearest
// first; the nearest is returned, the others are deferred.
// Nodes below `n_top` are read from the workgroup's LDS copy `top` (7 x 16 B
// per node) instead of through the vector memory pipeline.
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4f_t lds_v4f_t;
__device__ inline float4 lds_f4(const float4* p) {
  v4f_t v = *(const lds_v4f_t*)p;
  return make_float4(v.x, v.y, v.z, v.w);
}
constexpr int kTopF4 = 7;
template <bool SORT = true>
__device__ inline int node_step(const DevScene& S, int cur, V3 o, V3 inv, float tmax, Stack& st, uint32_t lane,
                                unsigned long long* overflow, const float4* top = nullptr, int n_top = 0) {
  float4 lx, ly, lz, hx, hy, hz, rf;
  if (cur < n_top) {
    const float4* n = top + cur * kTopF4;
    lx = lds_f4(n); ly = lds_f4(n + 1); lz = lds_f4(n + 2); hx = lds_f4(n + 3); hy = lds_f4(n + 4); hz = lds_f4(n + 5);
    rf = lds_f4(n + 6);
  } else {
    const float4* n = S.nodes + (size_t)cur * kNodeF4;
    lx = n[0]; ly = n[1]; lz = n[2]; hx = n[3]; hy = n[4]; hz = n[5]; rf = n[6];
  }
  int r0 = __float_as_int(rf.x), r1 = __float_as_int(rf.y), r2 = __float_as_int(rf.z), r3 = __float_as_int(rf.w);
  float k0, k1, k2, k3;
  bool h0 = box_test(v3(lx.x, ly.x, lz.x), v3(hx.x, hy.x, hz.x), o, inv, tmax, S.box_scale, k0) && r0 != kRefEmpty;
  bool h1 = box_test(v3(lx.y, ly.y, lz.y), v3(hx.y, hy.y, hz.y), o, inv, tmax, S.box_scale, k1) && r1 != kRefEmpty;
  bool h2 = box_test(v3(lx.z, ly.z, lz.z), v3(hx.z, hy.z, hz.z), o, inv, tmax, S.box_scale, k2) && r2 != kRefEmpty;
  bool h3 = box_test(v3(lx.w, ly.w, lz.w), v3(hx.w, hy.w, hz.w), o, inv, tmax, S.box_scale, k3) && r3 != kRefEmpty;
  int nh = (int)h0 + (int)h1 + (int)h2 + (int)h3;
  if (nh == 0) return stack_next(st, lane, tmax);
  k0 = h0 ? k0 : INFINITY; k1 = h1 ? k1 : INFINITY; k2 = h2 ? k2 : INFINITY; k3 = h3 ? k3 : INFINITY;
#define CASTER_CSWAP(ka, ra, kb, rb)        \
  if (kb < ka) {                             \
    float tk_ = ka; ka = kb; kb = tk_;       \
    int tr_ = ra; ra = rb; rb = tr_;         \
  }
  if (SORT) {
    CASTER_CSWAP(k0, r0, k1, r1)
    CASTER_CSWAP(k2, r2, k3, r3)
    CASTER_CSWAP(k0, r0, k2, r2)
    CASTER_CSWAP(k1, r1, k3, r3)
    CASTER_CSWAP(k1, r1, k2, r2)
  } else {
    // any-hit rays: children that are hit first, in no particular order (the network
    // sorts the hit flags; an exchange moves a miss behind a hit)
#define CASTER_CMOVE(ha, ka, ra, hb, kb, rb)              \
    if (!ha && hb) {                                       \
      float tk_ = ka; ka = kb; kb = tk_;                   \
      int tr_ = ra; ra = rb; rb = tr_;                     \
      ha = true; hb = false;                               \
    }
    CASTER_CMOVE(h0, k0, r0, h1, k1, r1)
    CASTER_CMOVE(h2, k2, r2, h3, k3, r3)
    CASTER_CMOVE(h0, k0, r0, h2, k2, r2)
    CASTER_CMOVE(h1, k1, r1, h3, k3, r3)
    CASTER_CMOVE(h1, k1, r1, h2, k2, r2)
#undef CASTER_CMOVE
  }
#undef CASTER_CSWAP
  bool ok = true;
  if (nh > 3) ok &= stack_push(st, lane, r3, k3);
  if (nh > 2) ok &= stack_push(st, lane, r2, k2);
  if (nh > 1) ok &= stack_push(st, lane, r1, k1);
  if (!ok) atomicAdd(overflow, 1ull);
  return r0;
}

// Tests the primitives of the leaf starting at `i`.  Among hits at equal
// distance the higher primitive index wins, as in the reference's in-order
// scan with its `t <= tmax` acceptance.  ANY: returns at the first hit.
template <bool ANY>
__device__ inline bool leaf_step(const DevScene& S, int i, V3 o, V3 d, float maxD, float tmin, Hit& hit) {
  bool last;
  do {
    float t, b1, b2;
    if (prim_test(S, i, o, d, maxD, tmin, hit.t, t, b1, b2, last)) {
      if (ANY) return true;
      if (t < hit.t || i > hit.prim) { hit.t = t; hit.prim = i; hit.b1 = b1; hit.b2 = b2; }
    }
    ++i;
  } while (!last);
  return false;
}

// Closest-hit (ANY=false) or any-hit (ANY=true) walk of one ray by one lane.
template <bool ANY>
__device__ inline bool traverse(const DevScene& S, V3 o, V3 d, float tmin, float tmax, Hit& hit, Stack& st,
                                unsigned long long* overflow) {
  const uint32_t lane = threadIdx.x & 63u;
  V3 inv = v3(1.f / d.x, 1.f / d.y, 1.f / d.z);
  float maxD = maxcomp(vabs(d));
  hit.prim = -1;
  hit.t = tmax;
  hit.b1 = hit.b2 = 0.f;
  st.sp = 0;
  int cur = S.n_nodes ? 0 : kRefEmpty;
  while (cur != kRefEmpty) {
    if (cur >= 0) {
      cur = node_step(S, cur, o, inv, hit.t, st, lane, overflow);
    } else {
      if (leaf_step<ANY>(S, ~cur, o, d, maxD, tmin, hit)) return true;
      cur = stack_next(st, lane, hit.t);
    }
  }
  return hit.prim >= 0;
}

// ----------------------------------------------------------------- kernels
struct LaunchCtx {
  DevScene S;
  PathState P;
  Stats* stats;
};

// Slot of (iteration k of the batch, pixel slot `local` of one iteration).  The
// batch is laid out tile by tile: the n_iter wavefronts (8x8 pixels of one
// iteration each) of a tile follow one another, so that a contiguous part of a
// queue -- what one XCD's wavefronts consume -- covers a band of the image
// rather than a few iterations of all of it, and that XCD's L2 sees the
// geometry of that band.
#ifndef CASTER_TILE_MAJOR
#define CASTER_TILE_MAJOR 1
#endif
__device__ inline uint32_t batch_slot(const DevScene& S, uint32_t k, uint32_t local, uint32_t n_iter) {
#if CASTER_TILE_MAJOR
  return (((local >> 6) * n_iter + k) << 6) | (local & 63u);
#else
  return k * (uint32_t)S.slots_per_iter + local;
#endif
}
__device__ inline void batch_slot_split(const DevScene& S, uint32_t slot, uint32_t n_iter, uint32_t& k, uint32_t& local) {
#if CASTER_TILE_MAJOR
  uint32_t w = slot >> 6, tile = w / n_iter;
  k = w - tile * n_iter;
  local = (tile << 6) | (slot & 63u);
#else
  k = slot / (uint32_t)S.slots_per_iter;
  local = slot - k * (uint32_t)S.slots_per_iter;
#endif
}

// Camera samples for iterations [first_iter, first_iter + n_iter): seeds the
// RNG, draws the four camera numbers (pathtracer.cu:881-890), writes the
// extension ray and fresh path state.  Queue index == path slot here.
__global__ __launch_bounds__(kWave) void k_generate(LaunchCtx L, RayQueue q, uint32_t first_iter, uint32_t n_iter,
                                                    int y0, int y1) {
  const DevScene& S = L.S;
  uint32_t total = (uint32_t)S.slots_per_iter * n_iter;
  for (uint32_t idx = blockIdx.x * kWave + threadIdx.x; idx < total; idx += gridDim.x * kWave) {
    uint32_t k, local;
    batch_slot_split(S, idx, n_iter, k, local);
    uint32_t px, py;
    bool live = slot_pixel(S, local, px, py) && (int)py >= y0 && (int)py < y1;
    uint32_t iter = first_iter + k;
    float4 o4 = make_float4(0, 0, 0, __int_as_float(-1)), d4 = make_float4(0, 0, 1, -1.f);
    uint32_t rng = 1u;
    if (live) {
      uint32_t pixel = px + py * (uint32_t)S.width;
      rng = rng_seed(pixel, iter);
      float offsetx = rng_uniform(rng) - 0.5f;
      float offsety = rng_uniform(rng) - 0.5f;
      // Bdpt() draws no lens sample and shoots through the centre of the aperture (pathtracer.cu:1403-1408)
      float a1 = S.integrator == 4 /*IT_BDPT*/ ? 0.f : rng_uniform(rng);
      float a2 = S.integrator == 4 ? 0.f : rng_uniform(rng);
      V3 orig, dir;
      primary_ray(S.cam, px + offsetx, py + offsety, a1, a2, orig, dir);
      o4 = f4(orig, __int_as_float(-1));
      d4 = f4(dir, INFINITY);
    }
    uint32_t flags = 0;
    if (S.integrator == 2 /*IT_VPT*/) flags |= (uint32_t)(S.cam.medium + 1) << kFMediumShift;
    q.o[idx] = o4;
    q.d[idx] = d4;  // tmax < 0 marks a slot outside the image: extend skips it
    q.slot[idx] = (int)idx;
    L.P.beta_rng[idx] = f4(v3(1.f, 1.f, 1.f), rng);
    L.P.li_flags[idx] = f4(v3(0.f, 0.f, 0.f), flags);
    L.P.beta_prev[idx] = make_float4(0, 0, 0, 0);
  }
}

__global__ void k_set_count(uint32_t* c, uint32_t v) { *c = v; }

// Traces every ray of a queue with persistent wavefronts.
//
// A wavefront claims rays a chunk at a time from the queue's work counter and
// keeps its 64 lanes fed: when kRefillIdle lanes have finished their rays the
// idle lanes take the next unclaimed ones, so a long ray does not hold 63 idle
// lanes hostage.  Between refills every step does the kind of work (a 4-wide
// node, or one primitive test) that more lanes are waiting for.
//   ANY=false: closest hit; (t, prim, b1, b2) goes into the queue entry.
//   ANY=true:  any hit; an occluded ray clears the pending light-sampled term
//              of its path.
#ifndef CASTER_SPECULATE
#define CASTER_SPECULATE 1
#endif
#ifndef CASTER_ANY_UNSORTED
#define CASTER_ANY_UNSORTED 1
#endif
#ifndef CASTER_TRACE_WAVES
#define CASTER_TRACE_WAVES 5
#endif
#ifdef CASTER_WALK_STATS
// tuning aid: wavefront-level step counts of k_trace, [ANY][counter]
__device__ unsigned long long g_walk_stats[2][8];
#define CASTER_WS(i, v) ws[i] += (unsigned long long)(v)
#else
#define CASTER_WS(i, v)
#endif
template <bool ANY>
__global__ __launch_bounds__(kWave * kTraceBlockWaves, CASTER_TRACE_WAVES) void k_trace(LaunchCtx L, RayQueue q, WalkAux aux) {
  __shared__ uint2 s_stack[kTraceBlockWaves][kLdsDepth][kWave];
  __shared__ float4 s_top[kTopNodes > 0 ? kTopNodes * kTopF4 : 1];
#ifdef CASTER_WALK_STATS
  unsigned long long ws[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  const DevScene& S = L.S;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave_in_block = threadIdx.x >> 6;
  const uint32_t wave_id = blockIdx.x * kTraceBlockWaves + wave_in_block;  // index of this wavefront in the grid
  const uint32_t count = *q.count;
  unsigned long long* overflow = &L.stats->stack_overflow;
  // the top of the tree, shared by the workgroup's wavefronts
  const int n_top = S.n_nodes < kTopNodes ? S.n_nodes : kTopNodes;
  if (kTopNodes > 0) {
    for (int i = (int)threadIdx.x; i < n_top * kTopF4; i += kWave * kTraceBlockWaves)
      s_top[i] = S.nodes[(size_t)(i / kTopF4) * kNodeF4 + (i % kTopF4)];
    __syncthreads();
  }
  Stack st{s_stack[wave_in_block], aux.spill + (wave_id * kWave + lane), aux.spill_stride, 0};
  int idx = -1;
  int cur = kRefEmpty;
  int pend = kRefEmpty;  // a leaf waiting for its primitives to be tested
  V3 o = v3(0, 0, 0), d = v3(0, 0, 1), inv = v3(0, 0, 1);
  const float tmin = S.epsilon;
  float maxD = 1.f;
  Hit hit;
  hit.t = 0.f; hit.prim = -1; hit.b1 = hit.b2 = 0.f;
  bool occluded = false;
  // this wavefront's first chunk is assigned statically (no atomic storm at
  // launch): the work counters start at the end of the static assignment
  const uint32_t part = ((count + kWorkShards - 1) / kWorkShards + kWave - 1) / kWave * kWave;
  // Rays are claimed kChunk at a time from a short queue (so that they spread over all wavefronts)
  // and up to four times as many at a time from a long one (fewer trips to the work counters).
  const uint32_t n_waves = gridDim.x * kTraceBlockWaves;
  const uint32_t per_wave = count / (n_waves * kChunk);
  const uint32_t chunk = per_wave >= 16 ? 4 * kChunk : per_wave >= 8 ? 2 * kChunk : kChunk;
  // the work counters count from the end of the statically assigned first chunk of every wavefront
  const uint32_t work_init = (n_waves + kWorkShards - 1) / kWorkShards * chunk;
  uint32_t shard = wave_id % kWorkShards, shards_tried = 0;
  uint32_t chunk_next, chunk_end;
  {
    uint32_t lo = shard * part, hi = lo + part < count ? lo + part : count;
    uint32_t base = (wave_id / kWorkShards) * chunk;
    chunk_next = lo + base < hi ? lo + base : 0;
    chunk_end = lo + base < hi ? (chunk_next + chunk < hi ? chunk_next + chunk : hi) : 0;
  }
  bool more = true;
  for (;;) {
    // retire finished rays
    if (idx >= 0 && cur == kRefEmpty && pend == kRefEmpty) {
      if (ANY) {
        if (occluded) {
          int slot = q.slot[idx];
          float4 px = L.P.pend_x[slot];
          px.x = 0.f; px.y = 0.f; px.z = 0.f;
          L.P.pend_x[slot] = px;
        }
      } else {
        q.hit[idx] = make_float4(hit.t, __int_as_float(hit.prim), hit.b1, hit.b2);
      }
      idx = -1;
    }
    // refill idle lanes
    unsigned long long idle_mask = __ballot(idx < 0);
    uint32_t n_idle = (uint32_t)__popcll(idle_mask);
    if (more && n_idle >= (uint32_t)kRefillIdle) {
      if (chunk_next == chunk_end) {
        // Claim the next chunk.  The queue is cut into kWorkShards contiguous
        // parts with a work counter each: same-address atomics with a return
        // value retire at only ~15 per microsecond on this chip, so one shared
        // counter would bound the kernel.  A wavefront works on "its" part
        // first and moves on to the next part when that one is exhausted.
        while (shards_tried < kWorkShards) {
          uint32_t lo = shard * part, hi = lo + part < count ? lo + part : count;
          // A part whose rays were all handed out statically needs no atomic to be found empty: in a
          // launch over a short queue that is every part, and 5120 wavefronts asking eight counters
          // in turn would take longer than the tracing (same-address atomics retire at ~20 per us).
          if (lo + work_init < hi) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(aux.work + shard * kWorkPad, chunk);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base) + work_init;
            if (lo + base < hi) {
              chunk_next = lo + base;
              chunk_end = chunk_next + chunk < hi ? chunk_next + chunk : hi;
              break;
            }
          }
          shard = shard + 1 == kWorkShards ? 0 : shard + 1;
          ++shards_tried;
        }
        if (shards_tried == kWorkShards) more = false;
      }
      if (more) {
        uint32_t avail = chunk_end - chunk_next;
        uint32_t give = avail < n_idle ? avail : n_idle;
        uint32_t prefix = (uint32_t)__popcll(idle_mask & ((1ull << lane) - 1ull));
        if (idx < 0 && prefix < give) {
          idx = (int)(chunk_next + prefix);
          float4 o4 = q.o[idx], d4 = q.d[idx];
          o = v3(o4); d = v3(d4);
          inv = v3(1.f / d.x, 1.f / d.y, 1.f / d.z);
          maxD = maxcomp(vabs(d));
          // every queued ray starts at the scene's epsilon; o.w carries a hit known
          // beforehand (a probe ray's nearest emitter, at distance tmax) or -1
          hit.t = d4.w; hit.prim = __float_as_int(o4.w); hit.b1 = hit.b2 = 0.f;
          occluded = false;
          st.sp = 0;
          // tmax < 0 marks a camera slot outside the image: nothing to trace
          cur = (d4.w >= 0.f && S.n_nodes) ? 0 : kRefEmpty;
        }
        chunk_next += give;
      }
    }
    CASTER_WS(6, 1);
    if (!__any(idx >= 0)) {
      if (!more) break;
      continue;
    }
    // Advance until enough lanes are free again.  Each step the wavefront does
    // whichever kind of work more of its lanes can do: one 4-wide node for every
    // lane that stands at an inner node, or one primitive test for every lane
    // that holds a leaf.  A lane that reaches a leaf parks it in `pend` (a leaf
    // reference is ~index of the next primitive to test, so the cursor lives in
    // the reference itself) and goes on with the next deferred node, so that it
    // keeps taking part in node steps; the parked leaf may later shorten the ray
    // and make some of that node work unnecessary, never wrong.
    for (;;) {
#if CASTER_SPECULATE
      if (cur < 0 && cur != kRefEmpty && pend == kRefEmpty) {
        pend = cur;
        cur = stack_next(st, lane, hit.t);
      }
      unsigned long long m_inner = __ballot(cur >= 0);
      unsigned long long m_leaf = __ballot(pend != kRefEmpty);
      unsigned long long m_busy = __ballot(cur != kRefEmpty || pend != kRefEmpty);
      int n_inner = __popcll(m_inner), n_leaf = __popcll(m_leaf);
      int n_free = kWave - __popcll(m_busy);
      if (m_busy == 0) break;
      if (more && n_free >= kRefillIdle) break;
      CASTER_WS(4, __popcll(m_busy));
      CASTER_WS(5, __popcll(m_inner & m_leaf));
      if (n_inner * CASTER_VOTE_NUM >= n_leaf * CASTER_VOTE_DEN) {
        CASTER_WS(0, 1);
        CASTER_WS(1, n_inner);
        if (cur >= 0) cur = node_step<!ANY || !CASTER_ANY_UNSORTED>(S, cur, o, inv, hit.t, st, lane, overflow, s_top, n_top);
      } else if (pend != kRefEmpty) {
        CASTER_WS(2, 1);
        CASTER_WS(3, n_leaf);
        int i = ~pend;
        float t, b1, b2;
        bool last;
        bool found = prim_test(S, i, o, d, maxD, tmin, hit.t, t, b1, b2, last);
        if (found) {
          if (ANY) {
            occluded = true;
          } else if (t < hit.t || i > hit.prim) {
            hit.t = t; hit.prim = i; hit.b1 = b1; hit.b2 = b2;
          }
        }
        if (ANY && found) {
          cur = kRefEmpty;
          pend = kRefEmpty;
        } else {
          pend = last ? kRefEmpty : ~(i + 1);
        }
      }
#else
      unsigned long long m_inner = __ballot(cur >= 0);
      unsigned long long m_leaf = __ballot(cur < 0 && cur != kRefEmpty);
      int n_inner = __popcll(m_inner), n_leaf = __popcll(m_leaf);
      int n_free = kWave - n_inner - n_leaf;
      if (n_inner + n_leaf == 0) break;
      if (more && n_free >= kRefillIdle) break;
      if (n_inner * CASTER_VOTE_NUM >= n_leaf * CASTER_VOTE_DEN) {
        if (cur >= 0) cur = node_step<!ANY || !CASTER_ANY_UNSORTED>(S, cur, o, inv, hit.t, st, lane, overflow, s_top, n_top);
      } else if (cur < 0 && cur != kRefEmpty) {
        int i = ~cur;
        float t, b1, b2;
        bool last;
        bool found = prim_test(S, i, o, d, maxD, tmin, hit.t, t, b1, b2, last);
        if (found) {
          if (ANY) {
            occluded = true;
          } else if (t < hit.t || i > hit.prim) {
            hit.t = t; hit.prim = i; hit.b1 = b1; hit.b2 = b2;
          }
        }
        if (ANY && found) cur = kRefEmpty;
        else cur = last ? stack_next(st, lane, hit.t) : ~(i + 1);
      }
#endif
    }
  }
#ifdef CASTER_WALK_STATS
  if (lane == 0)
    for (int k = 0; k < 8; ++k) atomicAdd(&g_walk_stats[ANY ? 1 : 0][k], ws[k]);
#endif
}

// -------------------------------------------------------------- shading data
struct Isect {
  V3 pos, n, dpdu;
  V2 uv;
  int mat, light, med_in, med_out;
};

// Rebuilds the reference's Intersection record from (ray, hit): Triangle.h:
// 121-141 (shading normal = interpolated normal, tangent re-orthogonalised),
// sphere.h:69-88, line.h:74-84.
__device__ inline void load_isect(const DevScene& S, V3 o, V3 d, const Hit& h, Isect& is) {
  const float4* sh = S.shade + (size_t)h.prim * kShadeF4;
  // the primitive type sits in the shading record too, so that a shaded hit
  // touches one 128-byte line and not a second one in the intersection records
  uint32_t type = __float_as_uint(sh[7].x) & kPrimTypeMask;
  float4 ids = sh[6];
  is.mat = __float_as_int(ids.x);
  is.light = __float_as_int(ids.y);
  is.med_in = __float_as_int(ids.z);
  is.med_out = __float_as_int(ids.w);
  is.pos = o + h.t * d;
  if (type == 0u) {
    float4 s0 = sh[0], s1 = sh[1], s2 = sh[2], s3 = sh[3], s4 = sh[4], s5 = sh[5];
    float b1 = h.b1, b2 = h.b2;
    V3 ns = normalize(v3(s0) * (1.f - b1 - b2) + v3(s1) * b1 + v3(s2) * b2);
    V3 ss = normalize(v3(s3) * (1.f - b1 - b2) + v3(s4) * b1 + v3(s5) * b2);
    V3 ts = cross(ss, ns);
    if (length(ts) > 0.f) {
      ts = normalize(ts);
      ss = cross(ts, ns);
    }
    is.n = ns;
    is.dpdu = ss;
    float b0 = 1.f - b1 - b2;
    is.uv = V2{s0.w * b0 + s2.w * b1 + s4.w * b2, s1.w * b0 + s3.w * b1 + s5.w * b2};
  } else if (type == 2u) {
    float4 c = sh[0];
    is.n = normalize(is.pos - v3(c));
    V3 normal = is.n;
    float costheta = dot(normal, v3(0.f, 1.f, 0.f));
    float v = fps::acos_(costheta) * ONE_OVER_PI;
    float cosphi = dot(v3(1.f, 0.f, 0.f), v3(normal.x, 0.f, normal.z));
    float phi = fps::acos_(cosphi);
    phi = normal.z > 0.f ? TWOPI - phi : phi;
    float u = phi * ONE_OVER_TWO_PI;
    is.dpdu = normalize(v3(-TWOPI * is.pos.y, TWOPI * is.pos.x, 0));
    is.uv = V2{u, v};
    is.light = -1;
  } else {
    float4 a = sh[0], b = sh[1];
    float s = h.b1, d2 = h.b2;
    float r = a.w * (1 - s) + b.w * s;
    is.n = -d;
    is.uv = V2{s, sqrtf(d2) / r};
    V3 dpdu, dpdv;
    make_coordinate(is.n, dpdu, dpdv);
    is.dpdu = dpdu;
    is.light = -1;
    is.med_in = -1;
    is.med_out = -1;
  }
}

struct Mat {
  int type, tex;
  float alphaU, alphaV, insideIOR, outsideIOR;
  V3 k, eta, diffuse, specular;
};
__device__ inline void load_mat(const DevScene& S, int idx, Mat& m) {
  const float4* r = S.mats + (size_t)idx * kMatF4;
  float4 a = r[0], b = r[1];
  m.type = __float_as_int(a.x); m.alphaU = a.y; m.alphaV = a.z; m.insideIOR = a.w;
  m.outsideIOR = b.x; m.tex = __float_as_int(b.y);
  m.k = v3(r[2]); m.eta = v3(r[3]); m.diffuse = v3(r[4]); m.specular = v3(r[5]);
}

// getTexel / GetTexel (pathtracer.cu:333-368)
__device__ inline float4 texel(const uint32_t* pool, int w, int h, int x, int y) {
  float inv = 1.f / 255.f;
  float rx = (float)(x - (x / w) * w);
  float ry = (float)(y - (y / h) * h);
  x = fps::f2i_((rx < 0) ? rx + w : rx);
  y = fps::f2i_((ry < 0) ? ry + h : ry);
  if (x < 0) x = 0;
  if (x > w - 1) x = w - 1;
  if (y < 0) y = 0;
  if (y > h - 1) y = h - 1;
  uint32_t c = pool[(size_t)y * w + x];
  return make_float4((c & 255u) * inv, ((c >> 8) & 255u) * inv, ((c >> 16) & 255u) * inv, (c >> 24) * inv);
}
__device__ inline float4 sc4(float s, float4 a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }
__device__ inline float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ inline V3 albedo(const DevScene& S, const Mat& m, V2 uv) {
  if (m.tex == -1) return m.diffuse;
  const int* info = S.tex_info + 3 * m.tex;
  const uint32_t* pool = S.tex_pool + info[0];
  int w = info[1], h = info[2];
  float xx = w * uv.x;
  float yy = h * uv.y;
  int x = fps::f2i_(floorf(xx));
  int y = fps::f2i_(floorf(yy));
  float dx = fabsf(xx - x);
  float dy = fabsf(yy - y);
  float4 c00 = texel(pool, w, h, x, y), c10 = texel(pool, w, h, x + 1, y);
  float4 c01 = texel(pool, w, h, x, y + 1), c11 = texel(pool, w, h, x + 1, y + 1);
  float4 r = add4(sc4(1 - dy, add4(sc4(1 - dx, c00), sc4(dx, c10))), sc4(dy, add4(sc4(1 - dx, c01), sc4(dx, c11))));
  return v3(r.x, r.y, r.z);
}

// --------------------------------------------------------------------- BSDFs
// pathtracer.cu:45-178
// The diffuse texture is looked up once per vertex: the BSDF functions below are
// evaluated up to three times there (light sample, MIS sample, continuation)
// with the same uv, and each would repeat the four texel fetches.
__device__ inline void resolve_texture(const DevScene& S, Mat& m, V2 uv) {
  if (m.tex != -1 && (m.type == 0 || m.type == 5)) {  // the materials that read it: Lambertian, Substrate
    m.diffuse = albedo(S, m, uv);
    m.tex = -1;
  }
}
__device__ inline V3 spherical_dir(float s, float c, float phi) { return v3(s * fps::cos_(phi), s * fps::sin_(phi), c); }
__device__ inline float dielectric_fresnel(float cosi, float cost, float etai, float etat) {
  float Rparl = (etat * cosi - etai * cost) / (etat * cosi + etai * cost);
  float Rperp = (etai * cosi - etat * cost) / (etai * cosi + etat * cost);
  return (Rparl * Rparl + Rperp * Rperp) * 0.5f;
}
__device__ inline V3 conduct_fresnel(float cosi, V3 eta, V3 k) {
  V3 tmp = (eta * eta + k * k) * cosi * cosi;
  V3 Rparl2 = (tmp - eta * cosi * 2.f + 1.f) / (tmp + eta * cosi * 2.f + 1.f);
  V3 tmp_f = (eta * eta + k * k);
  V3 Rperp2 = (tmp_f - eta * cosi * 2.f + cosi * cosi) / (tmp_f + eta * cosi * 2.f + cosi * cosi);
  return (Rparl2 + Rperp2) * 0.5f;
}
__device__ inline float ggx_d(V3 wh, V3 normal, V3 dpdu, float alphaU, float alphaV) {
  float costheta = dot(wh, normal);
  if (costheta <= 0.f) return 0.f;
  costheta = fps::clamp_(costheta, 0.f, 1.f);
  float cos2Theta = costheta * costheta;
  float sin2Theta = 1.f - cos2Theta;
  float cos4Theta = cos2Theta * cos2Theta;
  float tantheta2 = sin2Theta / cos2Theta;
  V3 dir = normalize(wh - costheta * normal);
  float cosphi = dot(dir, dpdu);
  float e = tantheta2 * (cosphi * cosphi / (alphaU * alphaU) + (1.f - cosphi) / (alphaV * alphaV));
  return 1.f / (PI * alphaU * alphaV * cos4Theta * (1 + e) * (1 + e));
}
__device__ inline float smith_g(V3 w, V3 normal, V3 wh, V3 dpdu, float alphaU, float alphaV) {
  float wdn = dot(w, normal);
  if (wdn * dot(w, wh) < 0.f) return 0.f;
  float sinTheta = sqrtf(fps::clamp_(1.f - wdn * wdn, 0.f, 1.f));
  float tanTheta = sinTheta / wdn;
  if (isinf_(tanTheta)) return 0.f;
  V3 dir = normalize(w - wdn * normal);
  float cosPhi = dot(dir, dpdu);
  float cos2Phi = cosPhi * cosPhi;
  float sin2Phi = 1.f - cos2Phi;
  float alpha2 = cos2Phi * (alphaU * alphaU) + sin2Phi * (alphaV * alphaV);
  float alpha2Tan2Theta = alpha2 * tanTheta * tanTheta;
  return 2.f / (1.f + sqrtf(1 + alpha2Tan2Theta));
}
__device__ inline float ggx_g(V3 wo, V3 wi, V3 normal, V3 wh, V3 dpdu, float aU, float aV) {
  return smith_g(wo, normal, wh, dpdu, aU, aV) * smith_g(wi, normal, wh, dpdu, aU, aV);
}
__device__ inline V3 sample_ggx(float alphaU, float alphaV, float u1, float u2) {
  if (alphaU == alphaV) {
    float costheta = sqrtf((1.f - u1) / (u1 * (alphaU * alphaV - 1.f) + 1.f));
    float sintheta = sqrtf(1.f - costheta * costheta);
    float phi = TWOPI * u2;
    return spherical_dir(sintheta, costheta, phi);
  }
  float phi = fps::atan_(alphaV / alphaU * fps::tan_(TWOPI * u2 + .5f * PI));
  if (u2 > .5f) phi += PI;
  float sinphi = fps::sin_(phi), cosphi = fps::cos_(phi);
  const float alphaU2 = alphaU * alphaU;
  const float alpha2 = 1.0f / (cosphi * cosphi / alphaU2 + sinphi * sinphi / alphaU2);
  float tanTheta2 = alpha2 * u1 / (1.f - u1);
  float costheta = 1 / sqrtf(1 + tanTheta2);
  float sintheta = sqrtf(fps::fmax_(0.f, 1.f - costheta * costheta));
  return spherical_dir(sintheta, costheta, phi);
}
__device__ inline V3 reflect(V3 in, V3 nor) { return 2.f * dot(in, nor) * nor - in; }
__device__ inline V3 refract(V3 in, V3 nor, float etai, float etat) {
  float cosi = dot(in, nor);
  bool enter = cosi > 0;
  if (!enter) { float t = etai; etai = etat; etat = t; }
  float eta = etai / etat;
  float sini2 = 1.f - cosi * cosi;
  float sint2 = sini2 * eta * eta;
  float cost = sqrtf(1.f - sint2);
  return normalize((nor * cosi - in) * eta + (enter ? -cost : cost) * nor);
}
__device__ inline float pow5(float v) { return (v * v) * (v * v) * v; }
__device__ inline V3 schlick(V3 Rs, float c) { return Rs + pow5(1.f - c) * (1 - Rs); }
__device__ inline float power_heuristic(float fPdf, float gPdf) {
  float f = 1 * fPdf, g = 1 * gPdf;
  return (f * f) / (f * f + g * g);
}
__device__ inline float luminance(V3 c) { return dot(c, v3(0.212671f, 0.715160f, 0.072169f)); }
__device__ inline bool same_hemisphere(V3 in, V3 out, V3 nor) { return dot(in, nor) * dot(out, nor) > 0; }
__device__ inline bool is_delta(int t) { return t == 1 || t == 2; }  // mirror, dielectric

// SampleBSDF (pathtracer.cu:500-698); `radiance` false is TransportMode::Importance (light paths),
// which drops the radiance scaling of refraction (:549-550, :683-684).
__device__ void sample_bsdf(const DevScene& S, const Mat& m, V3 in, V3 nor, V2 uv, V3 dpdu, V3 u, V3& out, V3& fr,
                            float& pdf, bool radiance = true) {
  switch (m.type) {
    case 0: {  // lambertian
      V3 n = nor;
      if (dot(nor, in) < 0) n = -n;
      out = cosine_hemisphere(u.x, u.y, pdf);
      V3 uu = dpdu, ww = cross(uu, n);
      out = to_world(out, uu, ww, n);
      fr = albedo(S, m, uv) * ONE_OVER_PI;
      break;
    }
    case 1:  // mirror
      out = reflect(in, nor);
      fr = m.specular / fabsf(dot(out, nor));
      pdf = 1.f;
      break;
    case 2: {  // dielectric
      V3 wi = -in;
      V3 normal = nor;
      float ei = m.outsideIOR, et = m.insideIOR;
      float cosi = dot(wi, normal);
      if (cosi > 0) { float t = et; et = ei; ei = t; }
      float eta = ei / et;
      float sint2 = eta * eta * (1.f - cosi * cosi);
      float cost = sqrtf(1.f - sint2 < 0.f ? 0.f : 1.f - sint2);
      V3 rdir = reflect(-wi, normal);
      V3 tdir = refract(in, nor, m.outsideIOR, m.insideIOR);
      if (sint2 > 1.f) {
        out = rdir;
        fr = m.specular / fabsf(dot(out, normal));
        pdf = 1.f;
        return;
      }
      float fresnel = dielectric_fresnel(fabsf(cost), fabsf(cosi), et, ei);
      if (u.x > fresnel) {
        out = tdir;
        fr = m.specular / fabsf(dot(out, normal)) * (1.f - fresnel);
        if (radiance) fr = fr * (eta * eta);
        pdf = 1.f - fresnel;
      } else {
        out = rdir;
        fr = m.specular / fabsf(dot(out, normal)) * fresnel;
        pdf = fresnel;
      }
      break;
    }
    case 4: {  // rough conductor
      V3 n = nor;
      if (dot(nor, in) < 0) n = -n;
      V3 wh = sample_ggx(m.alphaU, m.alphaV, u.x, u.y);
      V3 uu = dpdu, ww = cross(uu, n);
      wh = to_world(wh, uu, ww, n);
      out = reflect(in, wh);
      if (!same_hemisphere(in, out, nor)) { fr = v3(0, 0, 0); pdf = 0.f; return; }
      float cosi = dot(out, wh);
      V3 F = conduct_fresnel(fabsf(cosi), m.eta, m.k);
      float D = ggx_d(wh, n, dpdu, m.alphaU, m.alphaV);
      float G = ggx_g(in, out, n, wh, dpdu, m.alphaU, m.alphaV);
      fr = m.specular * F * D * G / (4.f * fabsf(dot(in, n)) * fabsf(dot(out, n)));
      pdf = D * fabsf(dot(wh, n)) / (4.f * fabsf(dot(in, wh)));
      break;
    }
    case 5: {  // substrate
      V3 n = nor;
      if (dot(nor, in) < 0) n = -n;
      if (u.x < 0.5f) {
        u.x = fps::fmin_(2.f * u.x, 0x1.fffffep-1f);
        out = cosine_hemisphere(u.x, u.y, pdf);
        V3 uu = dpdu, ww = cross(uu, n);
        out = to_world(out, uu, ww, n);
      } else {
        u.x = fps::fmin_(2 * (u.x - 0.5f), 0x1.fffffep-1f);
        V3 wh = sample_ggx(m.alphaU, m.alphaV, u.x, u.y);
        V3 uu = dpdu, ww = cross(uu, n);
        wh = to_world(wh, uu, ww, n);
        out = reflect(in, wh);
      }
      if (!same_hemisphere(in, out, n)) { fr = v3(0, 0, 0); pdf = 0.f; return; }
      float cosThetaWo = fabsf(dot(in, n));
      float cosThetaWi = fabsf(dot(out, n));
      V3 Rd = albedo(S, m, uv);
      V3 Rs = m.specular;
      V3 diffuse = (28.f / (23.f * PI)) * Rd * (v3(1.f, 1.f, 1.f) - Rs) * (1 - pow5(1 - .5f * cosThetaWi)) *
                   (1 - pow5(1 - .5f * cosThetaWo));
      V3 wh = normalize(in + out);
      float D = ggx_d(wh, n, dpdu, m.alphaU, m.alphaV);
      float mx = cosThetaWi > cosThetaWo ? cosThetaWi : cosThetaWo;
      V3 specular = D / (4.f * fabsf(dot(out, wh)) * mx) * schlick(Rs, dot(out, wh));
      fr = diffuse + specular;
      pdf = 0.5f * (fabsf(dot(out, n)) * ONE_OVER_PI + D * fabsf(dot(wh, n)) / (4.f * dot(in, wh)));
      break;
    }
    case 3: {  // rough dielectric
      V3 wo = -in;
      V3 n = nor;
      V3 wh = sample_ggx(m.alphaU, m.alphaV, u.x, u.y);
      V3 uu = dpdu, ww = cross(uu, n);
      wh = to_world(wh, uu, ww, n);
      float ei = m.outsideIOR, et = m.insideIOR;
      float cosi = dot(wo, n);
      if (cosi > 0) { float t = et; et = ei; ei = t; }
      float eta = ei / et;
      float D = ggx_d(wh, n, dpdu, m.alphaU, m.alphaV);
      float cosThetaI = dot(wo, wh);
      float sin2ThetaI = fps::fmax_(0.f, 1.f - cosThetaI * cosThetaI);
      float sin2ThetaT = eta * eta * sin2ThetaI;
      float cosThetaT = sqrtf(1.f - sin2ThetaT < 0.f ? 0.f : 1.f - sin2ThetaT);
      V3 rdir = reflect(-wo, wh);
      V3 tdir = normalize((wo - wh * cosThetaI) * eta + (cosi < 0 ? -cosThetaT : cosThetaT) * wh);
      if (sin2ThetaT > 1.f) {
        out = rdir;
        float G = ggx_g(in, out, n, wh, dpdu, m.alphaU, m.alphaV);
        fr = m.specular * D * G / (4.f * fabsf(dot(in, n)) * fabsf(dot(out, n)));
        pdf = D * fabsf(dot(wh, n)) / (4.f * fabsf(dot(wh, in)));
        return;
      }
      float fresnel = dielectric_fresnel(fabsf(cosThetaI), fabsf(cosThetaT), ei, et);
      if (u.z > fresnel) {
        out = tdir;
        float G = ggx_g(in, out, n, wh, dpdu, m.alphaU, m.alphaV);
        float c = et * dot(out, wh) + ei * dot(in, wh);
        fr = m.specular * ei * ei * D * G * (1.f - fresnel) * fabsf(dot(in, wh)) * fabsf(dot(out, wh)) /
             (fabsf(dot(out, n)) * fabsf(dot(in, n)) * c * c);
        if (radiance) fr = fr * (1.f / (eta * eta));
        pdf = (1.f - fresnel) * D * fabsf(dot(wh, n)) * et * et * fabsf(dot(out, wh)) / (c * c);
      } else {
        out = rdir;
        float G = ggx_g(in, out, n, wh, dpdu, m.alphaU, m.alphaV);
        fr = m.specular * fresnel * D * G / (4.f * fabsf(dot(in, n)) * fabsf(dot(out, n)));
        pdf = D * fabsf(dot(wh, n)) / (4.f * fabsf(dot(wh, in))) * fresnel;
      }
      break;
    }
    default:
      out = v3(0, 0, 1); fr = v3(0, 0, 0); pdf = 0.f;
  }
}

// Fr (pathtracer.cu:700-818)
__device__ void eval_bsdf(const DevScene& S, const Mat& m, V3 in, V3 out, V3 nor, V2 uv, V3 dpdu, V3& fr, float& pdf) {
  switch (m.type) {
    case 0:
      if (!same_hemisphere(in, out, nor)) { fr = v3(0, 0, 0); pdf = 0.f; return; }
      fr = albedo(S, m, uv) * ONE_OVER_PI;
      pdf = fabsf(dot(out, nor)) * ONE_OVER_PI;
      break;
    case 4: {
      if (!same_hemisphere(in, out, nor)) { fr = v3(0, 0, 0); pdf = 0; return; }
      V3 n = nor;
      if (dot(nor, in) < 0) n = -n;
      V3 wh = normalize(in + out);
      float cosi = dot(out, wh);
      float D = ggx_d(wh, n, dpdu, m.alphaU, m.alphaV);
      float G = ggx_g(in, out, n, wh, dpdu, m.alphaU, m.alphaV);
      V3 F = conduct_fresnel(fabsf(cosi), m.eta, m.k);
      fr = m.specular * F * D * G / (4.f * fabsf(dot(in, n)) * fabsf(dot(out, n)));
      pdf = D * fabsf(dot(wh, n)) / (4.f * fabsf(dot(in, wh)));
      break;
    }
    case 5: {
      if (!same_hemisphere(in, out, nor)) { fr = v3(0, 0, 0); pdf = 0; return; }
      V3 n = nor;
      if (dot(nor, in) < 0) n = -n;
      float cosThetaWo = fabsf(dot(in, n));
      float cosThetaWi = fabsf(dot(out, n));
      V3 Rd = albedo(S, m, uv);
      V3 Rs = m.specular;
      V3 diffuse = (28.f / (23.f * PI)) * Rd * (v3(1.f, 1.f, 1.f) - Rs) * (1 - pow5(1 - .5f * cosThetaWi)) *
                   (1 - pow5(1 - .5f * cosThetaWo));
      V3 wh = normalize(in + out);
      float D = ggx_d(wh, n, dpdu, m.alphaU, m.alphaV);
      float mx = cosThetaWi > cosThetaWo ? cosThetaWi : cosThetaWo;
      V3 specular = D / (4.f * fabsf(dot(out, wh)) * mx) * schlick(Rs, dot(out, wh));
      fr = diffuse + specular;
      pdf = 0.5f * (fabsf(dot(out, n)) * ONE_OVER_PI + D * fabsf(dot(wh, n)) / (4.f * dot(in, wh)));
      break;
    }
    case 3: {
      V3 wo = -in;
      V3 n = nor;
      bool refl = dot(in, n) * dot(out, n) > 0;
      float ei = m.outsideIOR, et = m.insideIOR;
      float cosi = dot(wo, n);
      if (cosi > 0) { float t = et; et = ei; ei = t; }
      V3 wh = normalize(-(ei * in + et * out));
      float eta = ei / et;
      float D = ggx_d(wh, n, dpdu, m.alphaU, m.alphaV);
      float cosThetaI = dot(wo, wh);
      float sin2ThetaI = fps::fmax_(0.f, 1.f - cosThetaI * cosThetaI);
      float sin2ThetaT = eta * eta * sin2ThetaI;
      float cosThetaT = sqrtf(1.f - sin2ThetaT < 0.f ? 0.f : 1.f - sin2ThetaT);
      float fresnel = dielectric_fresnel(fabsf(cosThetaI), fabsf(cosThetaT), ei, et);
      if (!refl) {
        float G = ggx_g(in, out, n, wh, dpdu, m.alphaU, m.alphaV);
        float c = et * dot(out, wh) + ei * dot(in, wh);
        fr = m.specular * ei * ei * D * G * (1.f - fresnel) * fabsf(dot(in, wh)) * fabsf(dot(out, wh)) /
             (fabsf(dot(out, n)) * fabsf(dot(in, n)) * c * c);
        fr = fr * (1.f / (eta * eta));
        pdf = (1.f - fresnel) * D * fabsf(dot(wh, n)) * et * et * fabsf(dot(out, wh)) / (c * c);
      } else {
        float G = ggx_g(in, out, n, wh, dpdu, m.alphaU, m.alphaV);
        fr = m.specular * fresnel * D * G / (4.f * fabsf(dot(in, n)) * fabsf(dot(out, n)));
        pdf = fresnel * D * fabsf(dot(wh, n)) / (4.f * fabsf(dot(wh, in)));
      }
      break;
    }
    default:  // mirror, dielectric
      fr = v3(0, 0, 0);
      pdf = 0.f;
  }
}

// -------------------------------------------------------------------- lights
struct LightRec { V3 v0, v1, v2, n0, n1, n2, radiance; float area; int prim; };
__device__ inline void load_light(const DevScene& S, int idx, LightRec& l) {
  const float4* r = S.lights + (size_t)idx * kLightF4;
  float4 a = r[0], b = r[1], c = r[2], d = r[3], e = r[4], f = r[5];
  l.v0 = v3(a); l.v1 = v3(b); l.v2 = v3(c); l.n0 = v3(d); l.n1 = v3(e); l.n2 = v3(f);
  l.area = a.w; l.radiance = v3(b.w, c.w, d.w); l.prim = __float_as_int(e.w);
}

// Loads whose address is the same for every lane of the wavefront, from tables
// that no kernel writes (lights, the light distribution): through the constant
// address space they become scalar loads, which bypass the per-lane vector
// memory pipeline and return into scalar registers.
typedef __attribute__((address_space(4))) const v4f_t const_v4f_t;
typedef __attribute__((address_space(4))) const float const_f32_t;
__device__ inline float4 uniform_load4(const float4* p) {
  v4f_t v = *(const_v4f_t*)p;
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ inline float uniform_load(const float* p) { return *(const_f32_t*)p; }

// Environment lookup (infinite.h:46-93)
__device__ inline V3 inf_texel(const DevInfinite& I, int x, int y) {
  int width = I.w, height = I.h;
  float rx = (float)(x - (x / width) * width);
  float ry = (float)(y - (y / height) * height);
  x = fps::f2i_((rx < 0) ? rx + width : rx);
  y = fps::f2i_((ry < 0) ? ry + height : ry);
  if (x < 0) x = 0;
  if (x > width - 1) x = width - 1;
  if (y < 0) y = 0;
  if (y > height - 1) y = height - 1;
  const float* p = I.data + ((size_t)y * width + x) * 3;
  return v3(p[0], p[1], p[2]);
}
__device__ inline V3 inf_le(const DevInfinite& I, V3 dir) {
  V3 u = v3(I.u), v = v3(I.v), w = v3(I.w_);
  float costheta = dot(dir, v);
  float theta = fps::acos_(costheta);
  V3 d = normalize(dir - costheta * v);
  float cosphi = dot(d, u);
  float phi = fps::acos_(cosphi);
  float c = dot(d, w);
  phi = c > 0 ? TWOPI - phi : phi;
  float uu = phi / TWOPI;
  float vv = theta / PI;
  float tu = 1.f - uu, tv = vv;
  float xx = I.w * tu;
  float yy = I.h * tv;
  int x = fps::f2i_(floorf(xx));
  int y = fps::f2i_(floorf(yy));
  float dx = fabsf(xx - x);
  float dy = fabsf(yy - y);
  V3 c00 = inf_texel(I, x, y), c10 = inf_texel(I, x + 1, y);
  V3 c01 = inf_texel(I, x, y + 1), c11 = inf_texel(I, x + 1, y + 1);
  return (1 - dy) * ((1 - dx) * c00 + dx * c10) + dy * ((1 - dx) * c01 + dx * c11);
}

// LookUpLightDistribution (pathtracer.cu:181-190); no matching bin -> last bin.
__device__ inline int pick_light(const DevScene& S, float u, float& pdf) {
  // first bin [cdf[i], cdf[i+1]] that holds u (pathtracer.cu:181-190); the last bin when none does
  int n = S.n_cdf;
  int last = n - 2;
  if (last < 0) { pdf = 0.f; return 0; }
  int idx = -1;
  float s_prev = uniform_load(S.light_cdf);
  for (int i = 0; i + 1 < n; ++i) {  // the bounds are uniform, the loads scalar: no lane waits on another
    float e = uniform_load(S.light_cdf + i + 1);
    if (idx < 0 && u >= s_prev && u <= e) { idx = i; pdf = e - s_prev; }
    s_prev = e;
  }
  if (idx >= 0) return idx;
  pdf = uniform_load(S.light_cdf + last + 1) - uniform_load(S.light_cdf + last);
  return last;
}
__device__ inline void sample_one_light(const DevScene& S, uint32_t& rng, V3 pos, V3& radiance, V3& sdir,
                                        float& stmax, float& lightPdf, float& choicePdf) {
  int idx = pick_light(S, rng_uniform(rng), choicePdf);
  float ux = rng_uniform(rng);
  float uy = rng_uniform(rng);
  if (S.n_lights == 0 && !S.inf.valid) {
    radiance = v3(0, 0, 0); lightPdf = 1.f; sdir = v3(0, 0, 1); stmax = 0.f;
    return;
  }
  if (idx != S.n_lights) {
    LightRec l;
    load_light(S, idx, l);
    float su1 = sqrtf(ux);
    V2 uv = V2{1.f - su1, uy * su1};
    V3 p = uv.x * l.v0 + uv.y * l.v1 + (1 - uv.x - uv.y) * l.v2;
    V3 normal = normalize(uv.x * l.n0 + uv.y * l.n1 + (1 - uv.x - uv.y) * l.n2);
    V3 dir = p - pos;
    float pdf = 1.f / (l.area * fabsf(dot(normal, normalize(dir)))) * dot(dir, dir);
    if (dot(normal, dir) >= 0.f) pdf = 0.f;
    lightPdf = pdf;
    radiance = pdf != 0.f ? l.radiance : v3(0.f, 0.f, 0.f);
    sdir = normalize(dir);
    stmax = sqrtf(dot(dir, dir) - S.epsilon);
  } else {
    float pdfW;
    V3 dir = uniform_sphere(ux, uy, pdfW);
    sdir = dir;
    stmax = 2.f * S.inf.radius - S.epsilon;
    lightPdf = pdfW;
    radiance = inf_le(S.inf, dir);
  }
}

// Queues a path writes to while being shaded.
struct ShadeQueues {
  RayQueue next;    // continuation rays (closest hit)
  RayQueue shadow;  // light-sample rays (any hit)
  RayQueue probe;   // BSDF-sampled direct-light rays (closest hit + weighting)
};

// Two-level queue compaction.  A shading wavefront appends the rays it emits,
// ballot-compacted, to a staging region of its own (no atomics: the cursor is
// a wave-uniform register), and when it has shaded all its paths it reserves
// its final range in the shared queue with ONE atomic per queue and copies the
// region there.  Appending straight to the shared queue costs one same-address
// atomic per 64 paths per queue, which is what bounded the shade stage before.
struct StageArea {
  RayQueue next, shadow, probe;  // o/d/slot arrays of gridDim.x * region_cap entries; count unused
  uint32_t region_cap;           // entries per wavefront region
};
__device__ inline void stage_push(const RayQueue& stage, uint32_t base, uint32_t& n, bool want, int slot, V3 o,
                                  int seed, V3 d, float tmax) {
  unsigned long long mask = __ballot(want);
  if (mask == 0) return;
  uint32_t lane = threadIdx.x & 63u;
  if (want) {
    uint32_t at = base + n + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    stage.o[at] = f4(o, __int_as_float(seed));
    stage.d[at] = f4(d, tmax);
    stage.slot[at] = slot;
  }
  n += (uint32_t)__popcll(mask);
}
__device__ inline void stage_flush(const RayQueue& stage, uint32_t base, uint32_t n, const RayQueue& fin) {
  n = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
  if (n == 0) return;
  uint32_t lane = threadIdx.x & 63u;
  uint32_t off = 0;
  if (lane == 0) off = atomicAdd(fin.count, n);
  off = (uint32_t)__builtin_amdgcn_readfirstlane((int)off);
  for (uint32_t i = lane; i < n; i += kWave) {
    fin.o[off + i] = stage.o[base + i];
    fin.d[off + i] = stage.d[base + i];
    fin.slot[off + i] = stage.slot[base + i];
  }
}
// The rounds (groups of 64 queue entries) wavefront `blockIdx.x` shades: a
// contiguous range, so that the compacted output keeps the input's order.
__device__ inline void wave_rounds(uint32_t count, uint32_t& r0, uint32_t& r1) {
  uint32_t rounds = (count + kWave - 1) / kWave;
  uint32_t per = (rounds + gridDim.x - 1) / gridDim.x;
  r0 = blockIdx.x * per;
  r1 = r0 + per < rounds ? r0 + per : rounds;
}

// Adds the two direct-light terms left pending by the previous vertex:
// Li += beta * (Ld_light + Ld_bsdf)  (pathtracer.cu:984).
__device__ inline V3 resolve_pending(const PathState& P, int slot, V3 Li, uint32_t flags) {
  V3 x = v3(P.pend_x[slot]);
  V3 y = (flags & kFPendingY) ? v3(P.pend_y[slot]) : v3(0.f, 0.f, 0.f);
  V3 bp = v3(P.beta_prev[slot]);
  V3 Ld = x + y;
  return Li + bp * Ld;
}

// One vertex of Path() (pathtracer.cu:897-1007) for every path whose
// extension ray has been traced.
#ifndef CASTER_SHADE_WAVES
#define CASTER_SHADE_WAVES 4
#endif
#ifndef CASTER_PROBE_CULL
#define CASTER_PROBE_CULL 1
#endif
constexpr int kProbeCullLights = 8;  // emitter triangles tested directly before a probe ray is sent
__global__ __launch_bounds__(kWave, CASTER_SHADE_WAVES) void k_shade_pt(LaunchCtx L, RayQueue cur, ShadeQueues out, StageArea stage) {
  const DevScene& S = L.S;
  const PathState& P = L.P;
  const uint32_t count = *cur.count;
  uint32_t r0, r1;
  wave_rounds(count, r0, r1);
  const uint32_t sbase = blockIdx.x * stage.region_cap;
  uint32_t n_next = 0, n_shadow = 0, n_probe = 0;
  for (uint32_t r = r0; r < r1; ++r) {
    uint32_t idx = r * kWave + threadIdx.x;
    bool active = idx < count;
    int slot = 0;
    bool want_next = false, want_shadow = false, want_probe = false;
    V3 no = v3(0, 0, 0), nd = v3(0, 0, 1), so = no, sd = nd, po = no, pd = nd;
    float stmax = 0.f, ptmax = INFINITY;
    int pseed = -1;
    if (active) {
      slot = cur.slot[idx];
      float4 o4 = cur.o[idx], d4 = cur.d[idx], h4 = cur.hit[idx];
      active = d4.w >= 0.f || __float_as_int(h4.y) >= 0;
      if (!active) {
        // slot outside the image: nothing to do, it never contributes
      }
    }
    if (active) {
      float4 o4 = cur.o[idx], d4 = cur.d[idx], h4 = cur.hit[idx];
      V3 ro = v3(o4), rd = v3(d4);
      Hit h;
      h.t = h4.x; h.prim = __float_as_int(h4.y); h.b1 = h4.z; h.b2 = h4.w;
      float4 br = P.beta_rng[slot], lf = P.li_flags[slot];
      V3 beta = v3(br), Li = v3(lf);
      uint32_t rng = __float_as_uint(br.w), flags = __float_as_uint(lf.w);
      int bounces = (int)(flags & kFBounceMask);
      bool specular = (flags & kFSpecular) != 0;
      if (flags & kFPending) {
        Li = resolve_pending(P, slot, Li, flags);
        flags &= ~(kFPending | kFPendingY);
      }
      bool alive = true;
      if (h.prim < 0) {
        if ((bounces == 0 || specular) && S.inf.valid) Li = Li + beta * inf_le(S.inf, rd);
        alive = false;
      }
      if (alive) {
        Isect is;
        load_isect(S, ro, rd, h, is);
        V3 pos = is.pos, nor = is.n, dpdu = is.dpdu;
        V2 uv = is.uv;
        Mat m;
        load_mat(S, is.mat, m);
        resolve_texture(S, m, uv);
        if ((bounces == 0 || specular) && is.light != -1) {
          // Area::Le (area.h:36-39)
          LightRec l;
          load_light(S, is.light, l);
          V3 le = dot(nor, -rd) > 0.f ? l.radiance : v3(0, 0, 0);
          Li = Li + beta * le;
          alive = false;
        }
        if (alive) {
          if (!is_delta(m.type)) {
            // light sampling
            V3 radiance, ldir;
            float ltmax, lightPdf, choicePdf;
            sample_one_light(S, rng, pos, radiance, ldir, ltmax, lightPdf, choicePdf);
            V3 x = v3(0, 0, 0);
            if (!is_black(radiance)) {
              V3 fr;
              float samplePdf;
              eval_bsdf(S, m, -rd, ldir, nor, uv, dpdu, fr, samplePdf);
              float weight = power_heuristic(lightPdf * choicePdf, samplePdf);
              x = weight * fr * radiance * fabsf(dot(nor, ldir)) / (lightPdf * choicePdf);
              // The reference traces the shadow ray before it evaluates the BSDF
              // (pathtracer.cu:933-940); when the term is exactly zero (light behind
              // the surface, zero BSDF) visibility cannot change Ld, so no ray is sent.
              want_shadow = !is_black(x);
              so = pos; sd = ldir; stmax = ltmax;
            }
            // BSDF sampling for direct light
            V3 us;
            us.x = rng_uniform(rng); us.y = rng_uniform(rng); us.z = rng_uniform(rng);
            V3 mo, mfr;
            float mpdf;
            sample_bsdf(S, m, -rd, nor, uv, dpdu, us, mo, mfr, mpdf);
            float4 py = make_float4(0, 0, 0, 0);
            float absdot = 0.f;
            if (!is_black(mfr) && mpdf != 0) {
              want_probe = true;
              po = pos; pd = mo;
              py = f4(mfr, mpdf);
              absdot = fabsf(dot(mo, nor));
#if CASTER_PROBE_CULL
              // The probe only contributes when its closest hit is an emitter (no
              // environment light here), so a ray that cannot reach one is not
              // sent.  With few emitters their triangles are tested with the very
              // test the traversal uses: if none is hit, the closest hit cannot be
              // one; otherwise the ray starts out holding the nearest emitter hit,
              // so that the traversal only looks for something at least as close.
              if (!S.inf.valid) {
                // first the padded box around all emitters: a ray that misses it cannot hit one
                float tn;
                bool reach = S.n_lights > 0 &&
                             box_test(v3(S.emit_lo), v3(S.emit_hi), pos, v3(1.f / mo.x, 1.f / mo.y, 1.f / mo.z), INFINITY,
                                      S.box_scale, tn);
                if (!reach) {
                  want_probe = false;
                  py = make_float4(0, 0, 0, 0);
                } else if (S.n_lights <= kProbeCullLights) {
                  float pmaxD = maxcomp(vabs(mo));
                  for (int li = 0; li < S.n_lights; ++li) {
                    int lprim = __float_as_int(uniform_load4(S.lights + (size_t)li * kLightF4 + 4).w);
                    const float4* rec = S.isect + (size_t)lprim * kIsectF4;
                    float t, b1, b2;
                    bool last;
                    if (prim_test_rec(S, uniform_load4(rec), uniform_load4(rec + 1), uniform_load4(rec + 2), pos, mo, pmaxD,
                                      S.epsilon, INFINITY, t, b1, b2, last) &&
                        (t < ptmax || (t == ptmax && lprim > pseed) || pseed < 0)) {
                      ptmax = t;
                      pseed = lprim;
                    }
                  }
                  if (pseed < 0) { want_probe = false; py = make_float4(0, 0, 0, 0); }
                }
              }
#endif
            }
            // Li += beta * (x + y) is owed once visibility and the probe are known.
            // Without a probe y counts as zero and is not stored; when x is zero
            // as well the sum adds beta * 0, which only matters for a non-finite
            // beta (the sample then turns NaN and is dropped), so a finite beta
            // stores nothing at all.
            bool finite_beta = fabsf(beta.x) < INFINITY && fabsf(beta.y) < INFINITY && fabsf(beta.z) < INFINITY;
            if (want_probe || !is_black(x) || !finite_beta) {
              P.pend_x[slot] = f4(x, absdot);
              if (want_probe) P.pend_y[slot] = py;
              P.beta_prev[slot] = f4(beta, 0.f);
              flags |= kFPending | (want_probe ? kFPendingY : 0u);
            }
          }
          // continuation
          V3 u;
          u.x = rng_uniform(rng); u.y = rng_uniform(rng); u.z = rng_uniform(rng);
          V3 wo2, fr2;
          float pdf2;
          sample_bsdf(S, m, -rd, nor, uv, dpdu, u, wo2, fr2, pdf2);
          if (is_black(fr2)) {
            alive = false;
          } else {
            beta = beta * (fr2 * fabsf(dot(nor, wo2)) / pdf2);
            specular = is_delta(m.type);
            no = pos; nd = wo2;
            if (bounces > 3) {
              float illumate = fps::clamp_(1.f - luminance(beta), 0.f, 1.f);
              if (rng_uniform(rng) < illumate) alive = false;
              else beta = beta / (1 - illumate);
            }
            if (bounces + 1 >= S.maxDepth) alive = false;
          }
        }
      }
      want_next = alive;
      flags = (flags & ~(kFBounceMask | kFSpecular)) | (uint32_t)(bounces + 1) | (specular ? kFSpecular : 0u);
      P.beta_rng[slot] = f4(beta, rng);
      P.li_flags[slot] = f4(Li, flags);
    }
    stage_push(stage.next, sbase, n_next, want_next, slot, no, -1, nd, INFINITY);
    stage_push(stage.shadow, sbase, n_shadow, want_shadow, slot, so, -1, sd, stmax);
    stage_push(stage.probe, sbase, n_probe, want_probe, slot, po, pseed, pd, ptmax);
  }
  __syncthreads();  // the copy below reads what other lanes of this wavefront staged
  stage_flush(stage.next, sbase, n_next, out.next);
  stage_flush(stage.shadow, sbase, n_shadow, out.shadow);
  stage_flush(stage.probe, sbase, n_probe, out.probe);
}

// BSDF-sampled direct light (pathtracer.cu:949-983), after the probe rays have
// been traced: an emitter hit (or the environment on a miss) contributes
// weight * fr * Le * |cos| / pdf into the path's pending term.
__global__ __launch_bounds__(kWave) void k_probe_resolve(LaunchCtx L, RayQueue q) {
  const DevScene& S = L.S;
  const uint32_t count = *q.count;
  for (uint32_t idx = blockIdx.x * kWave + threadIdx.x; idx < count; idx += gridDim.x * kWave) {
    float4 o4 = q.o[idx], d4 = q.d[idx], h4 = q.hit[idx];
    int slot = q.slot[idx];
    V3 pos = v3(o4), out = v3(d4);
    Hit h;
    h.t = h4.x; h.prim = __float_as_int(h4.y); h.b1 = h4.z; h.b2 = h4.w;
    bool found = h.prim >= 0;
    float4 py = L.P.pend_y[slot];
    V3 fr = v3(py);
    float pdf = py.w;
    float absdot = L.P.pend_x[slot].w;
    V3 y = v3(0, 0, 0);
    if (found) {
      int light = __float_as_int(S.shade[(size_t)h.prim * kShadeF4 + 6].y);
      uint32_t type = __float_as_uint(S.shade[(size_t)h.prim * kShadeF4 + 7].x) & kPrimTypeMask;
      if (type == 0u && light != -1) {
        LightRec l;
        load_light(S, light, l);
        // a hit the ray started out with carries no barycentrics: the same test gives them again
        {
          float t, b1, b2;
          bool last;
          if (prim_test(S, h.prim, pos, out, maxcomp(vabs(out)), S.epsilon, INFINITY, t, b1, b2, last)) {
            h.b1 = b1; h.b2 = b2;
          }
        }
        const float4* sh = S.shade + (size_t)h.prim * kShadeF4;
        V3 n = normalize(v3(sh[0]) * (1.f - h.b1 - h.b2) + v3(sh[1]) * h.b1 + v3(sh[2]) * h.b2);
        V3 p = pos + h.t * out;
        V3 radiance = dot(n, -out) > 0.f ? l.radiance : v3(0, 0, 0);
        if (!is_black(radiance)) {
          float pdfA = 1.f / l.area;
          float choicePdf = S.light_cdf[light + 1] - S.light_cdf[light];
          float lenSquare = dot(p - pos, p - pos);
          float costheta = fabsf(dot(n, out));
          float lPdf = pdfA * lenSquare / (costheta);
          float weight = power_heuristic(pdf, lPdf * choicePdf);
          y = weight * fr * radiance * absdot / pdf;
        }
      }
    } else if (S.inf.valid) {
      V3 radiance = inf_le(S.inf, out);
      float choicePdf = S.light_cdf[S.n_lights + 1] - S.light_cdf[S.n_lights];
      float lightPdf = ONE_OVER_FOUR_PI;
      float weight = power_heuristic(pdf, lightPdf * choicePdf);
      y = weight * fr * radiance * absdot / pdf;
    }
    L.P.pend_y[slot] = f4(y, 0.f);
  }
}

// A vertex whose probe ray was never issued does not set kFPendingY, so its
// pend_y is not read; issued ones are always overwritten by k_probe_resolve.

// One vertex of Ao() (pathtracer.cu:841-868).
__global__ __launch_bounds__(kWave) void k_shade_ao(LaunchCtx L, RayQueue cur, ShadeQueues out, StageArea stage) {
  const DevScene& S = L.S;
  const PathState& P = L.P;
  const uint32_t count = *cur.count;
  uint32_t r0, r1;
  wave_rounds(count, r0, r1);
  const uint32_t sbase = blockIdx.x * stage.region_cap;
  uint32_t n_next = 0, n_shadow = 0, n_probe = 0;
  for (uint32_t r = r0; r < r1; ++r) {
    uint32_t idx = r * kWave + threadIdx.x;
    bool want_shadow = false;
    int slot = 0;
    V3 so = v3(0, 0, 0), sd = v3(0, 0, 1);
    if (idx < count) {
      slot = cur.slot[idx];
      float4 o4 = cur.o[idx], d4 = cur.d[idx], h4 = cur.hit[idx];
      if (d4.w >= 0.f) {
        V3 ro = v3(o4), rd = v3(d4);
        Hit h;
        h.t = h4.x; h.prim = __float_as_int(h4.y); h.b1 = h4.z; h.b2 = h4.w;
        float4 br = P.beta_rng[slot];
        uint32_t rng = __float_as_uint(br.w);
        uint32_t flags = 1u;
        V3 Li = v3(0, 0, 0);
        if (h.prim < 0) {
          Li = v3(1, 1, 1);
        } else {
          Isect is;
          load_isect(S, ro, rd, h, is);
          V3 n = is.n;
          float pdf = 0.f;
          if (dot(-rd, n) < 0.f) n = -n;
          float u1 = rng_uniform(rng);
          float u2 = rng_uniform(rng);
          V3 wi = cosine_hemisphere(u1, u2, pdf);
          V3 s = is.dpdu;
          V3 t = cross(n, s);
          wi = to_world(wi, s, t, n);
          float cosine = dot(wi, n);
          float v = cosine / pdf;
          P.pend_x[slot] = f4(v3(v, v, v), 0.f);
          P.beta_prev[slot] = f4(v3(1, 1, 1), 0.f);
          flags |= kFPending;
          want_shadow = true;
          so = is.pos; sd = wi;
        }
        P.beta_rng[slot] = f4(v3(br), rng);
        P.li_flags[slot] = f4(Li, flags);
      }
    }
    stage_push(stage.shadow, sbase, n_shadow, want_shadow, slot, so, -1, sd, S.maxDist);
  }
  __syncthreads();
  stage_flush(stage.shadow, sbase, n_shadow, out.shadow);
  (void)n_next; (void)n_probe;
}

// ---- volumetric path tracing ------------------------------------------------
__device__ inline V3 vexp(V3 c) { return v3(fps::exp_(c.x), fps::exp_(c.y), fps::exp_(c.z)); }
struct MediumRec { V3 sigmaS, sigmaT; float g; int type; float invMax; };
__device__ inline void load_medium(const DevScene& S, int idx, MediumRec& m) {
  const float4* r = S.media + (size_t)idx * kMediumF4;
  float4 a = r[0], b = r[1], c = r[2];
  m.g = a.w; m.sigmaS = v3(b); m.type = __float_as_int(b.w); m.sigmaT = v3(c); m.invMax = c.w;
}
// Density grid of a heterogeneous medium (medium.h:41-49).
struct GridRec { int nx, ny, nz, off, iterMax; V3 p0, p1; };
__device__ inline void load_grid(const DevScene& S, int idx, GridRec& g) {
  const float4* r = S.media + (size_t)idx * kMediumF4;
  float4 a = r[0], d = r[3], e = r[4];
  g.nx = __float_as_int(a.x); g.ny = __float_as_int(a.y); g.nz = __float_as_int(a.z);
  g.p0 = v3(d); g.off = __float_as_int(d.w); g.p1 = v3(e); g.iterMax = __float_as_int(e.w);
}
__device__ inline float het_cell(const DevScene& S, const GridRec& g, V3 p) {  // medium.h:124-129
  int x = fps::f2i_(p.x), y = fps::f2i_(p.y), z = fps::f2i_(p.z);
  if (x < 0 || x > g.nx - 1 || y < 0 || y > g.ny - 1 || z < 0 || z > g.nz - 1) return 0.f;
  return S.density[(size_t)g.off + (size_t)z * g.ny * g.nx + (size_t)y * g.nx + x];
}
__device__ inline float lerp_(float a, float b, float t) { return a + t * (b - a); }
__device__ inline float het_density(const DevScene& S, const GridRec& g, V3 p) {  // medium.h:111-122
  V3 ps = v3(p.x * g.nx - .5f, p.y * g.ny - .5f, p.z * g.nz - .5f);
  V3 psi = v3(floorf(ps.x), floorf(ps.y), floorf(ps.z));
  V3 delta = ps - psi;
  float d00 = lerp_(het_cell(S, g, psi), het_cell(S, g, psi + v3(1, 0, 0)), delta.x);
  float d10 = lerp_(het_cell(S, g, psi + v3(0, 1, 0)), het_cell(S, g, psi + v3(1, 1, 0)), delta.x);
  float d01 = lerp_(het_cell(S, g, psi + v3(0, 0, 1)), het_cell(S, g, psi + v3(1, 0, 1)), delta.x);
  float d11 = lerp_(het_cell(S, g, psi + v3(0, 1, 1)), het_cell(S, g, psi + v3(1, 1, 1)), delta.x);
  float d0 = lerp_(d00, d10, delta.y);
  float d1 = lerp_(d01, d11, delta.y);
  return lerp_(d0, d1, delta.z);
}
// Medium transmittance over [0, tmax] of the ray (o, d): Homogeneous::Tr
// (medium.h:13-16) or ratio tracking through the density grid (medium.h:53-79),
// which draws random numbers.
__device__ V3 medium_tr(const DevScene& S, int medium, V3 o, V3 d, float tmax, uint32_t& rng) {
  MediumRec m;
  load_medium(S, medium, m);
  if (m.type == 0) return vexp(m.sigmaT * (-tmax));
  GridRec g;
  load_grid(S, medium, g);
  float sigma = dot(m.sigmaT, v3(0.212671f, 0.715160f, 0.072169f));
  V3 ext = g.p1 - g.p0;
  float tr = 1.f;
  float dist = 0.f;
  int iter = g.iterMax;
  while (true) {
    dist += -fps::log_(rng_uniform(rng)) * m.invMax / sigma;
    if (dist >= tmax) break;
    V3 p = o + d * dist;
    p = (p - g.p0) / ext;
    tr *= 1.f - het_density(S, g, p) * m.invMax;
    if (tr < 0.1f) {
      float q = 1.f - tr;
      if (rng_uniform(rng) < q) return v3(0.f, 0.f, 0.f);
      tr /= (1.f - q);
    }
    if (--iter == 0) break;
  }
  return v3(tr, tr, tr);
}

constexpr int kMaxCrossings = 1024;

// Tr() (pathtracer.cu:307-331): transmittance along a shadow ray, marching
// through index-matched boundaries; an opaque hit gives zero.
__device__ V3 transmittance(const DevScene& S, V3 o, V3 d, float tmin, float tmax, int medium, Stack& st,
                            Stats* stats, uint32_t& segs, uint32_t& rng) {
  V3 tr = v3(1, 1, 1);
  float remaining = tmax;
  float seg_tmax = tmax;
  for (int guard = 0; guard < kMaxCrossings; ++guard) {
    Hit h;
    ++segs;
    bool found = traverse<false>(S, o, d, tmin, seg_tmax, h, st, &stats->stack_overflow);
    int mat = -1, med_in = -1, med_out = -1;
    V3 n = v3(0, 0, 0);
    if (found) {
      Isect is;
      load_isect(S, o, d, h, is);
      mat = is.mat; med_in = is.med_in; med_out = is.med_out; n = is.n;
      if (mat != -1) return v3(0, 0, 0);
    }
    float ray_tmax = found ? h.t : seg_tmax;
    if (medium != -1) tr = tr * medium_tr(S, medium, o, d, ray_tmax, rng);
    if (!found) break;
    int m = dot(d, n) > 0 ? med_out : med_in;
    remaining -= ray_tmax;
    o = o + ray_tmax * d;
    medium = m;
    tmin = S.epsilon;
    seg_tmax = remaining;
  }
  return tr;
}

// One event of Volpath() (pathtracer.cu:1040-1223).  Shadow transmittance and
// the BSDF-sampled light probe are walked inside this kernel; only the
// continuation ray goes back through the extend stage.
__global__ __launch_bounds__(kWave) void k_shade_vpt(LaunchCtx L, RayQueue cur, ShadeQueues out, StageArea stage, WalkAux aux) {
  __shared__ uint2 s_stack[kLdsDepth][kWave];
  Stack st{s_stack, aux.spill + (blockIdx.x * kWave + threadIdx.x), aux.spill_stride, 0};
  const DevScene& S = L.S;
  const PathState& P = L.P;
  const uint32_t count = *cur.count;
  uint32_t stat_tr = 0, stat_probe = 0;
  uint32_t r0, r1;
  wave_rounds(count, r0, r1);
  const uint32_t sbase = blockIdx.x * stage.region_cap;
  uint32_t n_next = 0, n_shadow = 0, n_probe = 0;
  for (uint32_t r = r0; r < r1; ++r) {
    uint32_t idx = r * kWave + threadIdx.x;
    bool want_next = false;
    int slot = 0;
    V3 no = v3(0, 0, 0), nd = v3(0, 0, 1);
    if (idx < count) {
      slot = cur.slot[idx];
      float4 o4 = cur.o[idx], d4 = cur.d[idx], h4 = cur.hit[idx];
      if (d4.w >= 0.f || __float_as_int(h4.y) >= 0) {
        V3 ro = v3(o4), rd = v3(d4);
        Hit h;
        h.t = h4.x; h.prim = __float_as_int(h4.y); h.b1 = h4.z; h.b2 = h4.w;
        float4 br = P.beta_rng[slot], lf = P.li_flags[slot];
        V3 beta = v3(br), Li = v3(lf);
        uint32_t rng = __float_as_uint(br.w), flags = __float_as_uint(lf.w);
        int bounces = (int)(flags & kFBounceMask);
        bool specular = (flags & kFSpecular) != 0;
        int rmed = (int)((flags & kFMediumMask) >> kFMediumShift) - 1;
        int crossings = (int)__float_as_uint(P.beta_prev[slot].w);
        int nmed = rmed;
        bool alive = true;
        bool count_bounce = true;
        if (h.prim < 0) {
          if ((bounces == 0 || specular) && S.inf.valid) Li = Li + beta * inf_le(S.inf, rd);
          alive = false;
        } else {
          Isect is;
          load_isect(S, ro, rd, h, is);
          V3 pos = is.pos, nor = is.n, dpdu = is.dpdu;
          V2 uv = is.uv;
          float rtmax = h.t;  // the reference's r.tmax after Intersect
          float sampledDist = 0.f;
          bool sampledMedium = false;
          MediumRec med;
          if (rmed != -1) {
            load_medium(S, rmed, med);
            if (med.type == 0) {
              // Homogeneous::Sample (medium.h:18-38)
              int channel = fps::f2i_(rng_uniform(rng) * 3);
              channel = channel == 3 ? 2 : channel;
              float sT = channel == 0 ? med.sigmaT.x : (channel == 1 ? med.sigmaT.y : med.sigmaT.z);
              float dist = -fps::log_(1 - rng_uniform(rng)) / sT;
              dist = dist < rtmax ? dist : rtmax;
              sampledMedium = dist < rtmax;
              if (sampledMedium) sampledDist = dist;
              V3 Tr = vexp(med.sigmaT * -dist);
              V3 density = sampledMedium ? (med.sigmaT * Tr) : Tr;
              float pdf = 0;
              pdf += density.x; pdf += density.y; pdf += density.z;
              pdf /= 3.f;
              if (pdf == 0) pdf = 1;
              beta = beta * (sampledMedium ? (Tr * med.sigmaS / pdf) : (Tr / pdf));
            } else {
              // Heterogeneous::Sample (medium.h:81-109): delta tracking
              GridRec g;
              load_grid(S, rmed, g);
              float sigma = dot(med.sigmaT, v3(0.212671f, 0.715160f, 0.072169f));
              V3 ext = g.p1 - g.p0;
              float dist = 0.f;
              int iter = g.iterMax;
              while (true) {
                dist += -fps::log_(rng_uniform(rng)) * med.invMax / sigma;
                if (dist >= rtmax) break;
                V3 p = ro + rd * dist;
                p = (p - g.p0) / ext;
                if (het_density(S, g, p) * med.invMax > rng_uniform(rng)) {
                  sampledDist = dist;
                  sampledMedium = true;
                  beta = beta * (med.sigmaS / med.sigmaT);
                  break;
                }
                if (--iter == 0) break;
              }
              if (!sampledMedium) beta = beta * v3(1.f, 1.f, 1.f);
            }
          }
          if (is_black(beta)) {
            alive = false;
          } else if (sampledMedium) {
            V3 samplePos = ro + sampledDist * rd;
            V3 radiance, ldir;
            float ltmax, lightPdf, choicePdf;
            sample_one_light(S, rng, samplePos, radiance, ldir, ltmax, lightPdf, choicePdf);
            // the reference walks Tr() whatever the light sample is (pathtracer.cu:1079); the walk
            // draws random numbers only inside a density grid, so without one it is skipped
            V3 tr = (is_black(radiance) && !S.has_hetero)
                        ? v3(1, 1, 1)
                        : transmittance(S, samplePos, ldir, S.epsilon, ltmax, rmed, st, L.stats, stat_tr, rng);
            // Medium::Phase / SamplePhase (medium.h:146-186)
            float phase;
            if (med.g == 0) phase = ONE_OVER_FOUR_PI;
            else {
              float costheta = dot(-rd, ldir);
              float cubicTerm = (1.f + med.g * med.g - 2.f * med.g * costheta);
              phase = ONE_OVER_FOUR_PI * (1.f - med.g * med.g) / sqrtf(cubicTerm * cubicTerm * cubicTerm);
            }
            if (!is_black(radiance)) Li = Li + tr * beta * phase * radiance / (lightPdf * choicePdf);
            float ux = rng_uniform(rng);
            float uy = rng_uniform(rng);
            V3 dir;
            if (med.g == 0) {
              float pdf;
              dir = uniform_sphere(ux, uy, pdf);
            } else {
              float g = med.g, costheta;
              if (fabsf(g) < 1e-3f) costheta = 1.f - 2.f * ux;
              else {
                float sqrtTerm = (1.f - g * g) / (1.f - g + 2.f * g * ux);
                costheta = (1.f + g * g - sqrtTerm * sqrtTerm) / (2.f * g);
              }
              float sintheta = sqrtf(1.f - costheta * costheta);
              float phi = TWOPI * uy;
              float sinphi = fps::sin_(phi), cosphi = fps::cos_(phi);
              dir = v3(sintheta * cosphi, sintheta * sinphi, costheta);
            }
            no = samplePos; nd = dir; nmed = rmed;
            specular = false;
          } else {
            LightRec l;
            bool emitter = (bounces == 0 || specular) && is.light != -1;
            if (emitter) {
              V3 tr = v3(1.f, 1.f, 1.f);
              if (rmed != -1) tr = medium_tr(S, rmed, ro, rd, rtmax, rng);
              load_light(S, is.light, l);
              V3 le = dot(nor, -rd) > 0.f ? l.radiance : v3(0, 0, 0);
              Li = Li + tr * beta * le;
              alive = false;
            } else if (is.mat == -1) {
              if (++crossings > kMaxCrossings) alive = false;
              count_bounce = false;
              nmed = dot(rd, is.n) > 0 ? is.med_out : is.med_in;
              no = pos; nd = rd;
            } else {
              Mat m;
              load_mat(S, is.mat, m);
              resolve_texture(S, m, uv);
              if (!is_delta(m.type)) {
                V3 Ld = v3(0.f, 0.f, 0.f);
                V3 radiance, ldir;
                float ltmax, lightPdf, choicePdf;
                sample_one_light(S, rng, pos, radiance, ldir, ltmax, lightPdf, choicePdf);
                if (!is_black(radiance)) {
                  V3 fr;
                  float samplePdf;
                  eval_bsdf(S, m, -rd, ldir, nor, uv, dpdu, fr, samplePdf);
                  V3 tr = transmittance(S, pos, ldir, S.epsilon, ltmax, rmed, st, L.stats, stat_tr, rng);
                  float weight = power_heuristic(lightPdf * choicePdf, samplePdf);
                  Ld = Ld + weight * tr * fr * radiance * fabsf(dot(nor, ldir)) / (lightPdf * choicePdf);
                }
                V3 us;
                us.x = rng_uniform(rng); us.y = rng_uniform(rng); us.z = rng_uniform(rng);
                V3 mo, mfr;
                float mpdf;
                sample_bsdf(S, m, -rd, nor, uv, dpdu, us, mo, mfr, mpdf);
                if (!is_black(mfr) && mpdf != 0) {
                  Hit lh;
                  ++stat_probe;
                  bool found = traverse<false>(S, pos, mo, S.epsilon, INFINITY, lh, st, &L.stats->stack_overflow);
                  if (found) {
                    int light = __float_as_int(S.shade[(size_t)lh.prim * kShadeF4 + 6].y);
                    uint32_t type = __float_as_uint(S.shade[(size_t)lh.prim * kShadeF4 + 7].x) & kPrimTypeMask;
                    if (type == 0u && light != -1) {
                      load_light(S, light, l);
                      const float4* sh = S.shade + (size_t)lh.prim * kShadeF4;
                      V3 n = normalize(v3(sh[0]) * (1.f - lh.b1 - lh.b2) + v3(sh[1]) * lh.b1 + v3(sh[2]) * lh.b2);
                      V3 p = pos + lh.t * mo;
                      V3 rad2 = dot(n, -mo) > 0.f ? l.radiance : v3(0, 0, 0);
                      if (!is_black(rad2)) {
                        float pdfA = 1.f / l.area;
                        float cp = S.light_cdf[light + 1] - S.light_cdf[light];
                        float lenSquare = dot(p - pos, p - pos);
                        float costheta = fabsf(dot(n, mo));
                        float lPdf = pdfA * lenSquare / (costheta);
                        float weight = power_heuristic(mpdf, lPdf * cp);
                        V3 tr = v3(1.f, 1.f, 1.f);
                        if (rmed != -1) tr = medium_tr(S, rmed, pos, mo, lh.t, rng);
                        Ld = Ld + weight * tr * mfr * rad2 * fabsf(dot(mo, nor)) / mpdf;
                      }
                    }
                  } else if (S.inf.valid) {
                    V3 rad2 = inf_le(S.inf, mo);
                    float cp = S.light_cdf[S.n_lights + 1] - S.light_cdf[S.n_lights];
                    float weight = power_heuristic(mpdf, ONE_OVER_FOUR_PI * cp);
                    V3 tr = v3(1.f, 1.f, 1.f);
                    if (rmed != -1) tr = medium_tr(S, rmed, pos, mo, INFINITY, rng);
                    Ld = Ld + weight * tr * mfr * rad2 * fabsf(dot(mo, nor)) / mpdf;
                  }
                }
                Li = Li + beta * Ld;
              }
              V3 u;
              u.x = rng_uniform(rng); u.y = rng_uniform(rng); u.z = rng_uniform(rng);
              V3 wo2, fr2;
              float pdf2;
              sample_bsdf(S, m, -rd, nor, uv, dpdu, u, wo2, fr2, pdf2);
              if (is_black(fr2)) {
                alive = false;
              } else {
                beta = beta * (fr2 * fabsf(dot(nor, wo2)) / pdf2);
                specular = is_delta(m.type);
                int mm = dot(wo2, nor) > 0 ? is.med_out : is.med_in;
                mm = dot(-rd, nor) * dot(wo2, nor) > 0 ? rmed : mm;
                nmed = mm;
                no = pos; nd = wo2;
              }
            }
          }
          if (alive && count_bounce) {
            if (bounces > 3) {
              float illumate = fps::clamp_(1.f - luminance(beta), 0.f, 1.f);
              if (rng_uniform(rng) < illumate) alive = false;
              else beta = beta / (1 - illumate);
            }
          }
          if (count_bounce) {
            if (bounces + 1 >= S.maxDepth) alive = false;
            bounces += 1;
          }
        }
        want_next = alive;
        flags = (uint32_t)bounces | (specular ? kFSpecular : 0u) | ((uint32_t)(nmed + 1) << kFMediumShift);
        P.beta_rng[slot] = f4(beta, rng);
        P.li_flags[slot] = f4(Li, flags);
        P.beta_prev[slot] = make_float4(0, 0, 0, __uint_as_float((uint32_t)crossings));
      }
    }
    stage_push(stage.next, sbase, n_next, want_next, slot, no, -1, nd, INFINITY);
  }
  __syncthreads();
  stage_flush(stage.next, sbase, n_next, out.next);
  (void)n_shadow; (void)n_probe;
  // ray statistics: one atomic per wavefront
  for (int off = 32; off > 0; off >>= 1) {
    stat_tr += (uint32_t)__shfl_down((int)stat_tr, off, 64);
    stat_probe += (uint32_t)__shfl_down((int)stat_probe, off, 64);
  }
  if ((threadIdx.x & 63u) == 0) {
    if (stat_tr) atomicAdd(&L.stats->rays[3], (unsigned long long)stat_tr);
    if (stat_probe) atomicAdd(&L.stats->rays[2], (unsigned long long)stat_probe);
  }
}

// ---- light tracing ------------------------------------------------------------
// LightTracing() (pathtracer.cu:1240-1375): paths start on the emitters and every vertex is
// connected to the pinhole and splatted onto the film.  The connections' transmittance walks run
// inside these kernels, lane by lane, as in k_shade_vpt; only the continuation ray goes back
// through the extend stage.  Splats are float atomics straight into the film sum, so the order
// in which a pixel's splats are added, and with it the last bits of the image, varies from run
// to run (as it does in the reference).

// Camera::SampleCamera (Camera.h:87-118).  pdf == 0: the pinhole does not see `pos`.
__device__ inline void sample_camera(const DevScene& S, V3 pos, V3& sdir, float& stmax, float& we, float& pdf,
                                     int& raster) {
  const DevCamera& c = S.cam;
  V3 dir = v3(c.pos) - pos;
  V3 ndir = normalize(dir);
  sdir = ndir;
  stmax = length(dir) - S.epsilon;
  V3 negative_dir = -ndir;
  V3 cndir = v3(dot(negative_dir, v3(c.u)), dot(negative_dir, v3(c.v)), dot(negative_dir, v3(c.w)));
  pdf = 0.f;
  if (cndir.z >= 0.f) return;
  float costheta = dot(cndir, v3(0, 0, -1.f));
  float scale = -c.distance / cndir.z;
  cndir = cndir * scale;
  V2 plane = V2{cndir.x / c.half_w, cndir.y / c.half_h};
  if (plane.x > 1.f || plane.x < -1.f || plane.y > 1.f || plane.y < -1.f) return;
  plane = V2{plane.x * 0.5f + 0.5f, plane.y * 0.5f + 0.5f};
  int x = fps::f2i_(floorf(plane.x * (c.res_x - 1) + 0.5f));
  int y = fps::f2i_(floorf(plane.y * (c.res_y - 1) + 0.5f));
  raster = fps::f2i_(y * c.res_x + x);
  pdf = dot(dir, dir) / costheta;
  we = c.distance * c.distance / (c.area * costheta * costheta * costheta * costheta);
}
__device__ inline void splat(const DevScene& S, float* acc, int raster, V3 v) {
  if (raster < 0 || raster >= S.width * S.height) return;  // a NaN position
  // adding a zero changes nothing (the sums start at +0), and a third to a half of all connections
  // are occluded or face away from the camera: their atomics are not issued
  if (v.x != 0.f) unsafeAtomicAdd(acc + 3 * (size_t)raster, v.x);
  if (v.y != 0.f) unsafeAtomicAdd(acc + 3 * (size_t)raster + 1, v.y);
  if (v.z != 0.f) unsafeAtomicAdd(acc + 3 * (size_t)raster + 2, v.z);
}
// Medium::Sample (medium.h:18-38 homogeneous, :81-109 delta tracking): scales beta, reports a
// scattering event before the surface at rtmax.
__device__ inline void medium_sample(const DevScene& S, int medium, const MediumRec& med, V3 ro, V3 rd, float rtmax,
                                     uint32_t& rng, V3& beta, float& sampledDist, bool& sampledMedium) {
  if (med.type == 0) {
    int channel = fps::f2i_(rng_uniform(rng) * 3);
    channel = channel == 3 ? 2 : channel;
    float sT = channel == 0 ? med.sigmaT.x : (channel == 1 ? med.sigmaT.y : med.sigmaT.z);
    float dist = -fps::log_(1 - rng_uniform(rng)) / sT;
    dist = dist < rtmax ? dist : rtmax;
    sampledMedium = dist < rtmax;
    if (sampledMedium) sampledDist = dist;
    V3 Tr = vexp(med.sigmaT * -dist);
    V3 density = sampledMedium ? (med.sigmaT * Tr) : Tr;
    float pdf = 0;
    pdf += density.x; pdf += density.y; pdf += density.z;
    pdf /= 3.f;
    if (pdf == 0) pdf = 1;
    beta = beta * (sampledMedium ? (Tr * med.sigmaS / pdf) : (Tr / pdf));
    return;
  }
  GridRec g;
  load_grid(S, medium, g);
  float sigma = dot(med.sigmaT, v3(0.212671f, 0.715160f, 0.072169f));
  V3 ext = g.p1 - g.p0;
  float dist = 0.f;
  int iter = g.iterMax;
  while (true) {
    dist += -fps::log_(rng_uniform(rng)) * med.invMax / sigma;
    if (dist >= rtmax) break;
    V3 p = ro + rd * dist;
    p = (p - g.p0) / ext;
    if (het_density(S, g, p) * med.invMax > rng_uniform(rng)) {
      sampledDist = dist;
      sampledMedium = true;
      beta = beta * (med.sigmaS / med.sigmaT);
      return;
    }
    if (--iter == 0) break;
  }
  beta = beta * v3(1.f, 1.f, 1.f);
}
// Medium::Phase and Medium::SamplePhase (medium.h:146-186)
__device__ inline float phase_eval(const MediumRec& med, V3 in, V3 out) {
  if (med.g == 0) return ONE_OVER_FOUR_PI;
  float costheta = dot(in, out);
  float cubicTerm = (1.f + med.g * med.g - 2.f * med.g * costheta);
  return ONE_OVER_FOUR_PI * (1.f - med.g * med.g) / sqrtf(cubicTerm * cubicTerm * cubicTerm);
}
__device__ inline V3 phase_sample(const MediumRec& med, float ux, float uy) {
  if (med.g == 0) {
    float pdf;
    return uniform_sphere(ux, uy, pdf);
  }
  float g = med.g, costheta;
  if (fabsf(g) < 1e-3f) costheta = 1.f - 2.f * ux;
  else {
    float sqrtTerm = (1.f - g * g) / (1.f - g + 2.f * g * ux);
    costheta = (1.f + g * g - sqrtTerm * sqrtTerm) / (2.f * g);
  }
  float sintheta = sqrtf(1.f - costheta * costheta);
  float phi = TWOPI * uy;
  float sinphi = fps::sin_(phi), cosphi = fps::cos_(phi);
  return v3(sintheta * cosphi, sintheta * sinphi, costheta);
}

// Start of every light path of iterations [first_iter, first_iter + n_iter) (pathtracer.cu:
// 1245-1272): seed, light choice, a point and a cosine-distributed direction on the chosen
// triangle (area.h:19-24, Triangle.h:157-166), and the connection of that point to the camera.
// A path that picks no emissive triangle (none in the scene, or the environment's bin, where the
// reference reads past its light array) draws its numbers and carries nothing.  `acc` null: no
// connection, only the rays (for the first-hit map).
// QUEUED (scenes without a density grid, where Tr() draws no random numbers): a connection is not
// walked here but sent through the traversal kernel -- its ray goes to `conn.shadow`, the path's
// slot to the list `conn.probe`, the operands of its contribution to pend_x / pend_y / pend_z, and
// k_lt_resolve splats it once k_tr_step / k_tr_finish have left its transmittance in pend_w.
struct LtConn {
  RayQueue shadow, probe;  // final queues
  StageArea stage;
};
enum { kLtSurface = 0, kLtMedium = 1, kLtStart = 2 };
template <bool QUEUED>
__global__ __launch_bounds__(kWave) void k_generate_lt(LaunchCtx L, RayQueue q, uint32_t first_iter, uint32_t n_iter,
                                                       int y0, int y1, float* acc, WalkAux aux, LtConn conn) {
  __shared__ uint2 s_stack[kLdsDepth][kWave];
  Stack st{s_stack, aux.spill + (blockIdx.x * kWave + threadIdx.x), aux.spill_stride, 0};
  const DevScene& S = L.S;
  uint32_t stat_tr = 0;
  uint32_t total = (uint32_t)S.slots_per_iter * n_iter;  // a multiple of the wavefront size
  const uint32_t sbase = blockIdx.x * conn.stage.region_cap;
  uint32_t n_conn = 0, n_list = 0;
  for (uint32_t idx = blockIdx.x * kWave + threadIdx.x; idx < total; idx += gridDim.x * kWave) {
    bool want_conn = false;
    V3 co = v3(0, 0, 0), cd = v3(0, 0, 1);
    float ctm = 0.f;
    uint32_t k, local;
    batch_slot_split(S, idx, n_iter, k, local);
    uint32_t px, py;
    bool live = slot_pixel(S, local, px, py) && (int)py >= y0 && (int)py < y1;
    uint32_t iter = first_iter + k;
    float4 o4 = make_float4(0, 0, 0, __int_as_float(-1)), d4 = make_float4(0, 0, 1, -1.f);
    uint32_t rng = 1u, flags = 0;
    V3 beta = v3(1.f, 1.f, 1.f);
    if (live) {
      uint32_t pixel = px + py * (uint32_t)S.width;
      rng = rng_seed(pixel, iter);
      float choicePdf;
      int li = pick_light(S, rng_uniform(rng), choicePdf);
      float ux = rng_uniform(rng);
      float uy = rng_uniform(rng);
      float uz = rng_uniform(rng);
      float uw = rng_uniform(rng);
      if (li < S.n_lights) {
        LightRec l;
        load_light(S, li, l);
        int lmed = __float_as_int(S.lights[(size_t)li * kLightF4 + 5].w);
        float su1 = sqrtf(ux);
        V2 uv = V2{1.f - su1, uy * su1};
        V3 pos = uv.x * l.v0 + uv.y * l.v1 + (1 - uv.x - uv.y) * l.v2;
        V3 nor = normalize(uv.x * l.n0 + uv.y * l.n1 + (1 - uv.x - uv.y) * l.n2);
        float pdfW;
        V3 dir = cosine_hemisphere(uz, uw, pdfW);
        V3 uu, ww;
        make_coordinate(nor, uu, ww);
        dir = to_world(dir, uu, ww, nor);
        float pdfA = 1.f / l.area;
        beta = beta * (l.radiance * fabsf(dot(dir, nor)) / (pdfA * pdfW * choicePdf));
        if (acc) {
          V3 cdir;
          float ctmax, we = 0.f, cameraPdf = 0.f;
          int raster = 0;
          sample_camera(S, pos, cdir, ctmax, we, cameraPdf, raster);
          if (cameraPdf != 0.f) {
            if (QUEUED) {
              want_conn = true;
              co = pos; cd = cdir; ctm = ctmax;
              L.P.pend_x[idx] = f4(l.radiance, 0.f);
              L.P.pend_z[idx] = make_float4(cameraPdf, __int_as_float(raster), __int_as_float(kLtStart), 0.f);
              L.P.pend_w[idx] = f4(v3(1.f, 1.f, 1.f), (uint32_t)lmed);
            } else {
              V3 tr = transmittance(S, pos, cdir, S.epsilon, ctmax, lmed, st, L.stats, stat_tr, rng);
              if (!is_black(tr)) splat(S, acc, raster, tr * l.radiance);
            }
          }
        }
        o4 = f4(pos, __int_as_float(-1));
        d4 = f4(dir, INFINITY);
        flags = (uint32_t)(lmed + 1) << kFMediumShift;
      }
    }
    q.o[idx] = o4;
    q.d[idx] = d4;
    q.slot[idx] = (int)idx;
    L.P.beta_rng[idx] = f4(beta, rng);
    L.P.li_flags[idx] = f4(v3(0.f, 0.f, 0.f), flags);
    L.P.beta_prev[idx] = make_float4(0, 0, 0, 0);
    if (QUEUED) {
      stage_push(conn.stage.shadow, sbase, n_conn, want_conn, (int)idx, co, -1, cd, ctm);
      stage_push(conn.stage.probe, sbase, n_list, want_conn, (int)idx, co, -1, cd, ctm);
    }
  }
  if (QUEUED) {
    __syncthreads();
    stage_flush(conn.stage.shadow, sbase, n_conn, conn.shadow);
    stage_flush(conn.stage.probe, sbase, n_list, conn.probe);
  }
  for (int off = 32; off > 0; off >>= 1) stat_tr += (uint32_t)__shfl_down((int)stat_tr, off, 64);
  if ((threadIdx.x & 63u) == 0 && stat_tr) atomicAdd(&L.stats->rays[3], (unsigned long long)stat_tr);
}

// One event of a light path (pathtracer.cu:1275-1374).
template <bool QUEUED>
__global__ __launch_bounds__(kWave) void k_shade_lt(LaunchCtx L, RayQueue cur, ShadeQueues out, StageArea stage,
                                                    WalkAux aux, float* acc) {
  __shared__ uint2 s_stack[kLdsDepth][kWave];
  Stack st{s_stack, aux.spill + (blockIdx.x * kWave + threadIdx.x), aux.spill_stride, 0};
  const DevScene& S = L.S;
  const PathState& P = L.P;
  const uint32_t count = *cur.count;
  uint32_t stat_tr = 0;
  uint32_t r0, r1;
  wave_rounds(count, r0, r1);
  const uint32_t sbase = blockIdx.x * stage.region_cap;
  uint32_t n_next = 0, n_conn = 0, n_list = 0;
  for (uint32_t r = r0; r < r1; ++r) {
    uint32_t idx = r * kWave + threadIdx.x;
    bool want_next = false, want_conn = false;
    int slot = 0;
    V3 no = v3(0, 0, 0), nd = v3(0, 0, 1), co = no, cd = nd;
    float ctm = 0.f;
    if (idx < count) {
      slot = cur.slot[idx];
      float4 o4 = cur.o[idx], d4 = cur.d[idx], h4 = cur.hit[idx];
      if (d4.w >= 0.f || __float_as_int(h4.y) >= 0) {
        V3 ro = v3(o4), rd = v3(d4);
        Hit h;
        h.t = h4.x; h.prim = __float_as_int(h4.y); h.b1 = h4.z; h.b2 = h4.w;
        float4 br = P.beta_rng[slot], lf = P.li_flags[slot];
        V3 beta = v3(br);
        uint32_t rng = __float_as_uint(br.w), flags = __float_as_uint(lf.w);
        int bounces = (int)(flags & kFBounceMask);
        int rmed = (int)((flags & kFMediumMask) >> kFMediumShift) - 1;
        int crossings = (int)__float_as_uint(P.beta_prev[slot].w);
        int nmed = rmed;
        bool alive = true;
        bool count_bounce = true;
        if (h.prim < 0) {
          alive = false;
        } else {
          Isect is;
          load_isect(S, ro, rd, h, is);
          V3 pos = is.pos, nor = is.n, dpdu = is.dpdu;
          V2 uv = is.uv;
          float rtmax = h.t;
          float sampledDist = 0.f;
          bool sampledMedium = false;
          MediumRec med;
          if (rmed != -1) {
            load_medium(S, rmed, med);
            medium_sample(S, rmed, med, ro, rd, rtmax, rng, beta, sampledDist, sampledMedium);
          }
          if (is_black(beta)) {
            alive = false;
          } else if (sampledMedium) {
            V3 samplePos = ro + sampledDist * rd;
            V3 cdir;
            float ctmax, we = 0.f, cameraPdf = 0.f;
            int raster = 0;
            sample_camera(S, samplePos, cdir, ctmax, we, cameraPdf, raster);
            // the reference walks Tr() whatever the camera sees (pathtracer.cu:1299); the walk draws
            // random numbers only inside a density grid, so without one a connection that cannot
            // contribute skips it
            if (QUEUED) {
              if (cameraPdf != 0.f) {
                want_conn = true;
                co = samplePos; cd = cdir; ctm = ctmax;
                P.pend_x[slot] = f4(beta, we);
                P.pend_y[slot] = make_float4(phase_eval(med, -rd, cdir), 0.f, 0.f, 0.f);
                P.pend_z[slot] = make_float4(cameraPdf, __int_as_float(raster), __int_as_float(kLtMedium), 0.f);
                P.pend_w[slot] = f4(v3(1.f, 1.f, 1.f), (uint32_t)rmed);
              }
            } else if (cameraPdf != 0.f || S.has_hetero) {
              V3 tr = transmittance(S, samplePos, cdir, S.epsilon, ctmax, rmed, st, L.stats, stat_tr, rng);
              float phase = phase_eval(med, -rd, cdir);
              if (cameraPdf != 0.f) {
                V3 Lc = beta * we * tr * phase / cameraPdf;
                if (!has_inf(Lc) && !has_nan(Lc)) splat(S, acc, raster, Lc);
              }
            }
            float ux = rng_uniform(rng);
            float uy = rng_uniform(rng);
            no = samplePos; nd = phase_sample(med, ux, uy); nmed = rmed;
          } else if (is.mat == -1) {
            if (++crossings > kMaxCrossings) alive = false;
            count_bounce = false;
            nmed = dot(rd, is.n) > 0 ? is.med_out : is.med_in;
            no = pos; nd = rd;
          } else {
            Mat m;
            load_mat(S, is.mat, m);
            resolve_texture(S, m, uv);
            if (!is_delta(m.type)) {
              V3 cdir;
              float ctmax, we = 0.f, cameraPdf = 0.f;
              int raster = 0;
              sample_camera(S, pos, cdir, ctmax, we, cameraPdf, raster);
              if (cameraPdf != 0.f) {
                V3 fr;
                float unuse;
                if (QUEUED) {
                  eval_bsdf(S, m, -rd, cdir, nor, uv, dpdu, fr, unuse);
                  want_conn = true;
                  co = pos; cd = cdir; ctm = ctmax;
                  P.pend_x[slot] = f4(beta, we);
                  P.pend_y[slot] = f4(fr, fabsf(dot(cdir, nor)));
                  P.pend_z[slot] = make_float4(cameraPdf, __int_as_float(raster), __int_as_float(kLtSurface), 0.f);
                  P.pend_w[slot] = f4(v3(1.f, 1.f, 1.f), (uint32_t)rmed);
                } else {
                  V3 tr = transmittance(S, pos, cdir, S.epsilon, ctmax, rmed, st, L.stats, stat_tr, rng);
                  eval_bsdf(S, m, -rd, cdir, nor, uv, dpdu, fr, unuse);
                  V3 Lc = tr * beta * fr * we * fabsf(dot(cdir, nor)) / cameraPdf;
                  if (!has_inf(Lc) && !has_nan(Lc)) splat(S, acc, raster, Lc);
                }
              }
            }
            V3 u;
            u.x = rng_uniform(rng); u.y = rng_uniform(rng); u.z = rng_uniform(rng);
            V3 wo2, fr2;
            float pdf2;
            sample_bsdf(S, m, -rd, nor, uv, dpdu, u, wo2, fr2, pdf2, false);
            if (is_black(fr2)) {
              alive = false;
            } else {
              beta = beta * (fr2 * fabsf(dot(wo2, nor)) / pdf2);
              int mm = dot(wo2, nor) > 0 ? is.med_out : is.med_in;
              mm = dot(-rd, nor) * dot(wo2, nor) > 0 ? rmed : mm;
              nmed = mm;
              no = pos; nd = wo2;
            }
          }
          if (alive && count_bounce) {
            if (bounces > 3) {
              float illumate = fps::clamp_(1.f - luminance(beta), 0.f, 1.f);
              if (rng_uniform(rng) < illumate) alive = false;
              else beta = beta / (1 - illumate);
            }
          }
          if (count_bounce) {
            if (bounces + 1 >= S.maxDepth) alive = false;
            bounces += 1;
          }
        }
        want_next = alive;
        flags = (uint32_t)bounces | ((uint32_t)(nmed + 1) << kFMediumShift);
        P.beta_rng[slot] = f4(beta, rng);
        P.li_flags[slot] = f4(v3(0.f, 0.f, 0.f), flags);
        P.beta_prev[slot] = make_float4(0, 0, 0, __uint_as_float((uint32_t)crossings));
      }
    }
    stage_push(stage.next, sbase, n_next, want_next, slot, no, -1, nd, INFINITY);
    if (QUEUED) {
      stage_push(stage.shadow, sbase, n_conn, want_conn, slot, co, -1, cd, ctm);
      stage_push(stage.probe, sbase, n_list, want_conn, slot, co, -1, cd, ctm);
    }
  }
  __syncthreads();
  stage_flush(stage.next, sbase, n_next, out.next);
  if (QUEUED) {
    stage_flush(stage.shadow, sbase, n_conn, out.shadow);
    stage_flush(stage.probe, sbase, n_list, out.probe);
  }
  for (int off = 32; off > 0; off >>= 1) stat_tr += (uint32_t)__shfl_down((int)stat_tr, off, 64);
  if ((threadIdx.x & 63u) == 0 && stat_tr) atomicAdd(&L.stats->rays[3], (unsigned long long)stat_tr);
}

// The connections of one round of light tracing on queues, their transmittance known: the same
// products, in the same order, as the in-kernel path above forms (pathtracer.cu:1271, 1303, 1343).
__global__ __launch_bounds__(kWave) void k_lt_resolve(LaunchCtx L, RayQueue list, float* acc) {
  const DevScene& S = L.S;
  const PathState& P = L.P;
  const uint32_t count = *list.count;
  for (uint32_t idx = blockIdx.x * kWave + threadIdx.x; idx < count; idx += gridDim.x * kWave) {
    int slot = list.slot[idx];
    float4 a = P.pend_x[slot], c = P.pend_z[slot];
    V3 tr = v3(P.pend_w[slot]);
    int raster = __float_as_int(c.y), kind = __float_as_int(c.z);
    float cameraPdf = c.x;
    if (kind == kLtStart) {
      if (!is_black(tr)) splat(S, acc, raster, tr * v3(a));
      continue;
    }
    float4 b = P.pend_y[slot];
    V3 beta = v3(a);
    float we = a.w;
    V3 Lc = kind == kLtMedium ? beta * we * tr * b.x / cameraPdf : tr * beta * v3(b) * we * b.w / cameraPdf;
    if (!has_inf(Lc) && !has_nan(Lc)) splat(S, acc, raster, Lc);
  }
}

// ---- bidirectional path tracing ---------------------------------------------------
// Bdpt() (pathtracer.cu:1379-1956): a camera sub-path and a light sub-path per sample, every
// prefix of the one joined to every prefix of the other, each join weighted by MisWeight().
// One sample per lane from seed to splat: the sample's random stream runs through both walks and
// then through the joins in (s, t) order (a join to a sampled light point draws three numbers,
// transmittance through a density grid draws more), so the joins of a sample cannot be issued
// before the walks are over nor reordered among themselves.  The reference keeps the two vertex
// arrays (2 x 67 x 100 bytes) in each thread's local memory; here they live in HBM, one column
// per lane of the grid, field by field (a wavefront's 64 lanes read 64 consecutive float4s).
// maxDepth is not used, as in the reference: sub-paths end by Russian roulette or after
// kBdptMaxDepth bounces.
constexpr int kBdptMaxDepth = 65;                   // BDPT_MAX_DEPTH (:1379)
constexpr int kBdptPathVerts = kBdptMaxDepth + 2;   // vertices per sub-path (:1930-1931)
constexpr int kBdptVertF4 = 5;                      // pos|fwd, n|rev, beta|mat, dpdu|light, uv|medium,delta,nd|q

struct BdptStore {
  float4* v;        // [field][vertex: camera path, then light path][lane of the grid]
  uint32_t stride;  // lanes of the grid
};
struct BVert {  // BdptVertex (:1381-1388)
  V3 pos; float fwd;
  V3 n; float rev;
  V3 beta; int mat;
  V3 dpdu; int light;
  V2 uv; int medium; int delta;
  int nd;  // neither this vertex nor the one before it on its sub-path is a delta vertex: MisWeight() counts it
};
// What MisWeight() needs of a vertex sits in one 8-byte word pair: (medium + 1 | delta << 16 | nd << 17) and
// q = remap(rev) / remap(fwd), the factor the vertex contributes to the running ratio whenever
// Connect() has not overridden its reverse density.  q is written together with rev.
__device__ inline int bv_pack(int medium, int delta, int nd) { return (medium + 1) | (delta << 16) | (nd << 17); }
__device__ inline float bv_remap(float pdf) { return pdf == 0 ? 1.f : pdf; }
__device__ inline size_t bv_at(const BdptStore& V, uint32_t g, int f, int v) {
  return (size_t)(f * (2 * kBdptPathVerts) + v) * V.stride + g;
}
__device__ inline void bv_store(const BdptStore& V, uint32_t g, int v, const BVert& x) {
  V.v[bv_at(V, g, 0, v)] = f4(x.pos, x.fwd);
  V.v[bv_at(V, g, 1, v)] = f4(x.n, x.rev);
  V.v[bv_at(V, g, 2, v)] = f4(x.beta, __int_as_float(x.mat));
  V.v[bv_at(V, g, 3, v)] = f4(x.dpdu, __int_as_float(x.light));
  V.v[bv_at(V, g, 4, v)] = make_float4(x.uv.x, x.uv.y, __int_as_float(bv_pack(x.medium, x.delta, x.nd)), 0.f);
}
__device__ inline void bv_load(const BdptStore& V, uint32_t g, int v, BVert& x) {
  float4 a = V.v[bv_at(V, g, 0, v)], b = V.v[bv_at(V, g, 1, v)], c = V.v[bv_at(V, g, 2, v)];
  float4 d = V.v[bv_at(V, g, 3, v)], e = V.v[bv_at(V, g, 4, v)];
  x.pos = v3(a); x.fwd = a.w; x.n = v3(b); x.rev = b.w; x.beta = v3(c); x.mat = __float_as_int(c.w);
  x.dpdu = v3(d); x.light = __float_as_int(d.w); x.uv = V2{e.x, e.y};
  int pk = __float_as_int(e.z);
  x.medium = (pk & 0xffff) - 1; x.delta = (pk >> 16) & 1; x.nd = (pk >> 17) & 1;
}
__device__ inline V3 bv_pos(const BdptStore& V, uint32_t g, int v) { return v3(V.v[bv_at(V, g, 0, v)]); }
__device__ inline float bv_fwd(const BdptStore& V, uint32_t g, int v) { return ((const float*)&V.v[bv_at(V, g, 0, v)])[3]; }
__device__ inline float bv_rev(const BdptStore& V, uint32_t g, int v) { return ((const float*)&V.v[bv_at(V, g, 1, v)])[3]; }
// rev of vertex v, whose fwd is `fwd`
__device__ inline void bv_set_rev(const BdptStore& V, uint32_t g, int v, float rev, float fwd) {
  ((float*)&V.v[bv_at(V, g, 1, v)])[3] = rev;
  ((float*)&V.v[bv_at(V, g, 4, v)])[3] = bv_remap(rev) / bv_remap(fwd);
}
__device__ inline float2 bv_mis(const BdptStore& V, uint32_t g, int v) {  // (packed word, q)
  const float2* p = (const float2*)&V.v[bv_at(V, g, 4, v)];
  return p[1];
}

// ConvertPdf (:1391-1399): a solid-angle density at `prev` towards `cur` as an area density at `cur`
__device__ inline float convert_pdf(float pdf, V3 prev_pos, V3 cur_pos, V3 cur_n) {
  V3 dir = prev_pos - cur_pos;
  float square = dot(dir, dir);
  dir = normalize(dir);
  float ret = pdf / square;
  if (!is_black(cur_n)) ret *= fabsf(dot(dir, cur_n));
  return ret;
}
// Camera::PdfCamera (Camera.h:121-126), the solid-angle part
__device__ inline float pdf_camera_w(const DevCamera& c, V3 dir) {
  float costheta = dot(dir, -v3(c.w));
  return c.distance * c.distance / (c.area * costheta * costheta * costheta);
}
// Medium::SamplePhase with the value of the phase function (medium.h:146-172)
__device__ inline void phase_sample_value(const MediumRec& med, float ux, float uy, V3& dir, float& phase) {
  float g = med.g;
  if (g == 0) {
    float pdf;
    phase = ONE_OVER_FOUR_PI;
    dir = uniform_sphere(ux, uy, pdf);
    return;
  }
  float costheta;
  if (fabsf(g) < 1e-3f) costheta = 1.f - 2.f * ux;
  else {
    float sqrtTerm = (1.f - g * g) / (1.f - g + 2.f * g * ux);
    costheta = (1.f + g * g - sqrtTerm * sqrtTerm) / (2.f * g);
  }
  float sintheta = sqrtf(1.f - costheta * costheta);
  float phi = TWOPI * uy;
  float sinphi = fps::sin_(phi), cosphi = fps::cos_(phi);
  dir = v3(sintheta * cosphi, sintheta * sinphi, costheta);
  float cubicTerm = (1.f + g * g - 2.f * g * costheta);
  phase = ONE_OVER_FOUR_PI * (1.f - g * g) / (cubicTerm * sqrtf(cubicTerm));
}

// State of a sub-path walk between two iterations of the loop shared by GenerateCameraPath
// (:1431-1534) and GenerateLightPath (:1569-1671), which differ in the transport mode of
// SampleBSDF only.
struct BdptWalk {
  V3 ro, rd;             // the ray under way
  V3 beta;
  V3 prev_pos, prev_n;   // position and normal of the last stored vertex
  float prev_fwd;        // its forward density
  int prev_delta;        // and whether it is a delta vertex
  float forward;         // solid-angle density with which the ray's direction was chosen
  int rmed;              // medium of the ray
  int bounces, nVertex, crossings;
  uint32_t rng;
};
// One iteration of that loop for the ray's closest hit `h` (`found` false: the ray left the
// scene).  Stores at most one vertex into column g of V (sub-path starting at vertex `base`,
// which the caller has stored) and returns whether the walk goes on with the ray left in `w`.
__device__ __noinline__ bool bdpt_walk_step(const DevScene& S, const BdptStore& V, uint32_t g, int base, bool radiance, bool found,
                                            const Hit& h, BdptWalk& w) {
  if (!found) return false;
  Isect is;
  load_isect(S, w.ro, w.rd, h, is);
  V3 pos = is.pos, nor = is.n;
  float sampledDist = 0.f;
  bool sampledMedium = false;
  MediumRec med;
  if (w.rmed != -1) {
    load_medium(S, w.rmed, med);
    medium_sample(S, w.rmed, med, w.ro, w.rd, h.t, w.rng, w.beta, sampledDist, sampledMedium);
  }
  if (is_black(w.beta)) return false;
  if (sampledMedium) {
    V3 samplePos = w.ro + sampledDist * w.rd;
    float ux = rng_uniform(w.rng);
    float uy = rng_uniform(w.rng);
    V3 dir;
    float phase;
    phase_sample_value(med, ux, uy, dir, phase);
    BVert x;
    x.pos = samplePos; x.n = v3(0.f, 0.f, 0.f); x.beta = w.beta; x.mat = -1; x.light = -1;
    x.dpdu = v3(0.f, 0.f, 0.f); x.uv = V2{0.f, 0.f}; x.medium = w.rmed; x.delta = 0; x.nd = !w.prev_delta;
    x.fwd = convert_pdf(w.forward, w.prev_pos, samplePos, x.n);
    x.rev = 0.f;
    bv_store(V, g, base + w.bounces + 1, x);
    w.forward = phase;
    bv_set_rev(V, g, base + w.bounces, convert_pdf(w.forward, samplePos, w.prev_pos, w.prev_n), w.prev_fwd);
    w.nVertex++;
    w.ro = samplePos; w.rd = dir;
    w.prev_pos = samplePos; w.prev_n = x.n; w.prev_fwd = x.fwd; w.prev_delta = 0;
  } else {
    if (is.mat == -1) {
      if (++w.crossings > kMaxCrossings) return false;
      w.rmed = dot(w.rd, is.n) > 0 ? is.med_out : is.med_in;
      w.ro = pos;
      return true;  // a boundary crossing is no bounce and skips the roulette (:1479-1486)
    }
    Mat m;
    load_mat(S, is.mat, m);
    resolve_texture(S, m, is.uv);
    BVert x;
    x.pos = pos; x.n = nor; x.beta = w.beta; x.mat = is.mat; x.light = is.light; x.dpdu = is.dpdu; x.uv = is.uv;
    x.medium = w.rmed; x.delta = is_delta(m.type) ? 1 : 0; x.nd = !x.delta && !w.prev_delta;
    x.fwd = convert_pdf(w.forward, w.prev_pos, pos, nor);
    x.rev = 0.f;
    bv_store(V, g, base + w.bounces + 1, x);
    w.nVertex++;
    const float below_fwd = w.prev_fwd;  // of the vertex whose rev is set further down
    w.prev_fwd = x.fwd; w.prev_delta = x.delta;
    V3 u;
    u.x = rng_uniform(w.rng); u.y = rng_uniform(w.rng); u.z = rng_uniform(w.rng);
    V3 out, fr;
    float pdf;
    sample_bsdf(S, m, -w.rd, nor, is.uv, is.dpdu, u, out, fr, pdf, radiance);
    if (is_black(fr)) return false;
    w.beta = w.beta * (fr * fabsf(dot(out, nor)) / pdf);
    w.forward = is_delta(m.type) ? 0.f : pdf;
    {
      V3 unuse;
      float rpdf;
      eval_bsdf(S, m, out, -w.rd, nor, is.uv, is.dpdu, unuse, rpdf);
      bv_set_rev(V, g, base + w.bounces, convert_pdf(rpdf, pos, w.prev_pos, w.prev_n), below_fwd);
    }
    int mm = dot(out, nor) > 0 ? is.med_out : is.med_in;
    mm = dot(-w.rd, nor) * dot(out, nor) > 0 ? w.rmed : mm;
    w.rmed = mm;
    w.ro = pos; w.rd = out;
    w.prev_pos = pos; w.prev_n = nor;
  }
  if (w.bounces > 3) {
    float rrPdf = fps::clamp_(1.f - luminance(w.beta), 0.f, 1.f);
    if (rng_uniform(w.rng) < rrPdf) return false;
    w.beta = w.beta / (1 - rrPdf);
  }
  ++w.bounces;
  return w.bounces < kBdptMaxDepth;
}
// The whole walk on one lane; returns the number of vertices of the sub-path.
__device__ inline int bdpt_walk(const DevScene& S, Stack& st, Stats* stats, const BdptStore& V, uint32_t g, int base, V3 ro,
                                V3 rd, int rmed, V3 beta, float forward, bool radiance, V3 prev_pos, V3 prev_n, float prev_fwd,
                                uint32_t& rng, uint32_t& n_ext) {
  BdptWalk w;
  w.ro = ro; w.rd = rd; w.beta = beta; w.prev_pos = prev_pos; w.prev_n = prev_n; w.forward = forward; w.rmed = rmed;
  w.prev_fwd = prev_fwd; w.prev_delta = 0;
  w.bounces = 0; w.nVertex = 1; w.crossings = 0; w.rng = rng;
  for (;;) {
    Hit h;
    ++n_ext;
    bool found = traverse<false>(S, w.ro, w.rd, S.epsilon, INFINITY, h, st, &stats->stack_overflow);
    if (!bdpt_walk_step(S, V, g, base, radiance, found, h, w)) break;
  }
  rng = w.rng;
  return w.nVertex;
}

// MisWeight (:1676-1704).  Connect() overwrites the reverse densities of the joined vertices
// before calling it and restores them afterwards; here they are passed in instead: crev1 / crev2
// stand for cameraPath[s-1].rev / [s-2].rev, lrev1 / lrev2 for lightPath[t-1].rev / [t-2].rev, and
// for t == 1 (a freshly sampled light point) lrev1 / lfwd0 for lightPath[0].rev / .fwd.
__device__ __noinline__ float bdpt_mis(const BdptStore& V, uint32_t g, int s, int t, float crev1, float crev2, float lrev1,
                                        float lrev2, float lfwd0) {
  if (s + t == 2) return 1.f;
  const int LB = kBdptPathVerts;
  float sumW = 0.f;
  float ri = 1.f;
  for (int i = s - 1; i > 0; --i) {
    float2 pq = bv_mis(V, g, i);
    float q = pq.y;  // remap(rev) / remap(fwd) of the stored vertex
    if (i >= s - 2) q = bv_remap(i == s - 1 ? crev1 : crev2) / bv_remap(bv_fwd(V, g, i));
    ri *= q;
    if ((__float_as_int(pq.x) >> 17) & 1) sumW += ri;
  }
  ri = 1.f;
  for (int i = t - 1; i >= 0; --i) {
    float2 pq = bv_mis(V, g, LB + i);
    float q = pq.y;
    if (t == 1) q = bv_remap(lrev1) / bv_remap(lfwd0);
    else if (i >= t - 2) q = bv_remap(i == t - 1 ? lrev1 : lrev2) / bv_remap(bv_fwd(V, g, LB + i));
    ri *= q;
    if ((__float_as_int(pq.x) >> 17) & 1) sumW += ri;
  }
  return 1.f / (sumW + 1.f);
}

// Phase function or BSDF of a sub-path vertex with its density, as Connect() picks between them,
// for the pair of directions (in, out) and, for the density only, the reversed pair (out, in):
// Connect() asks for both at every vertex it joins (:1767/1785, :1820/1833, :1876-1896), and the
// material is fetched once for the two.
__device__ __noinline__ void bdpt_vertex_fr(const DevScene& S, bool use_medium, const BVert& v, V3 in, V3 out, V3& fr, float& pdf,
                                            float& rpdf) {
  if (use_medium) {
    MediumRec med;
    load_medium(S, v.medium, med);
    float phase = phase_eval(med, in, out);
    pdf = phase;
    fr = v3(phase, phase, phase);
    rpdf = phase_eval(med, out, in);
  } else {
    Mat m;
    load_mat(S, v.mat, m);
    resolve_texture(S, m, v.uv);
    eval_bsdf(S, m, in, out, v.n, v.uv, v.dpdu, fr, pdf);
    V3 unuse;
    eval_bsdf(S, m, out, in, v.n, v.uv, v.dpdu, unuse, rpdf);
  }
}

// Connect() (:1706-1909) for the camera prefix of s vertices and the light prefix of t.
__device__ V3 bdpt_connect(const DevScene& S, Stack& st, Stats* stats, const BdptStore& V, uint32_t g, int s, int t, int& raster,
                           uint32_t& rng, uint32_t& n_tr) {
  const V3 zero = v3(0.f, 0.f, 0.f);
  const int LB = kBdptPathVerts;
  const float eps = S.epsilon;
  if (t == 0) {  // the camera path hit an emitter (:1710-1735)
    BVert cur;
    bv_load(V, g, s - 1, cur);
    if (cur.light == -1) return zero;
    V3 prev_pos = bv_pos(V, g, s - 2);
    V3 prev_n = v3(V.v[bv_at(V, g, 1, s - 2)]);
    V3 dir = normalize(prev_pos - cur.pos);
    LightRec l;
    load_light(S, cur.light, l);
    V3 Lr = zero + cur.beta * (dot(cur.n, dir) > 0.f ? l.radiance : zero);
    if (is_black(Lr)) return Lr;
    float choicePdf = S.light_cdf[cur.light + 1] - S.light_cdf[cur.light];
    float pdfA = 1.f / l.area;
    float pdfW = fabsf(dot(dir, cur.n)) * ONE_OVER_PI;
    float mis = bdpt_mis(V, g, s, t, pdfA * choicePdf, convert_pdf(pdfW, cur.pos, prev_pos, prev_n), 0.f, 0.f, 0.f);
    return mis * Lr;
  }
  if (t == 1) {  // a freshly sampled point on a light joined to the camera path (:1736-1794)
    BVert cur;
    bv_load(V, g, s - 1, cur);
    V3 prev_pos = bv_pos(V, g, s - 2);
    V3 prev_n = v3(V.v[bv_at(V, g, 1, s - 2)]);
    V3 in = normalize(prev_pos - cur.pos);
    bool isMedium = cur.mat == -1;
    float choicePdf;
    int idx = pick_light(S, rng_uniform(rng), choicePdf);
    float ux = rng_uniform(rng);
    float uy = rng_uniform(rng);
    if (idx >= S.n_lights) return zero;  // not an emissive triangle: the reference reads past its light array
    LightRec l;
    load_light(S, idx, l);
    float su1 = sqrtf(ux);
    V2 uv = V2{1.f - su1, uy * su1};
    V3 p = uv.x * l.v0 + uv.y * l.v1 + (1 - uv.x - uv.y) * l.v2;
    V3 lightNor = normalize(uv.x * l.n0 + uv.y * l.n1 + (1 - uv.x - uv.y) * l.n2);
    V3 d = p - cur.pos;
    float lightPdf = 1.f / (l.area * fabsf(dot(lightNor, normalize(d)))) * dot(d, d);
    if (dot(lightNor, d) >= 0.f) lightPdf = 0.f;
    V3 radiance = lightPdf != 0.f ? l.radiance : zero;
    V3 sdir = normalize(d);
    float stmax = sqrtf(dot(d, d) - eps);
    V3 lightPos = cur.pos + (stmax + eps) * sdir;
    if (is_black(radiance)) return zero;
    if (!isMedium && cur.delta) return zero;
    V3 tr = transmittance(S, cur.pos, sdir, eps, stmax, cur.medium, st, stats, n_tr, rng);
    if (is_black(tr)) return zero;
    V3 fr;
    float nextPdf, pdf;
    bdpt_vertex_fr(S, isMedium, cur, in, sdir, fr, nextPdf, pdf);
    float G = isMedium ? 1.f : fabsf(dot(cur.n, sdir));
    V3 Lr = zero + cur.beta * tr * fr * radiance * G / (lightPdf * choicePdf);
    if (is_black(Lr)) return zero;
    float pdfA = 1.f / l.area;
    float pdfW = fabsf(dot(sdir, lightNor)) * ONE_OVER_PI;
    float next_fwd = pdfA * choicePdf;
    float next_rev = convert_pdf(nextPdf, cur.pos, lightPos, lightNor);
    float cur_rev = convert_pdf(pdfW, lightPos, cur.pos, cur.n);
    float prev_rev = convert_pdf(pdf, cur.pos, prev_pos, prev_n);
    float mis = bdpt_mis(V, g, s, t, cur_rev, prev_rev, next_rev, 0.f, next_fwd);
    return mis * Lr;
  }
  if (s == 1) {  // the light path joined to the pinhole (:1795-1841)
    BVert cur;
    bv_load(V, g, LB + t - 1, cur);
    V3 prev_pos = bv_pos(V, g, LB + t - 2);
    V3 prev_n = v3(V.v[bv_at(V, g, 1, LB + t - 2)]);
    V3 in = normalize(prev_pos - cur.pos);
    bool isMedium = cur.mat == -1;
    V3 sdir;
    float stmax, we = 0.f, cameraPdf = 0.f;
    sample_camera(S, cur.pos, sdir, stmax, we, cameraPdf, raster);
    if (cameraPdf == 0) return zero;
    if (!isMedium && cur.delta) return zero;
    V3 tr = transmittance(S, cur.pos, sdir, eps, stmax, cur.medium, st, stats, n_tr, rng);
    if (is_black(tr)) return zero;
    V3 fr;
    float nextPdf, pdf, costheta = fabsf(dot(sdir, cur.n));
    bdpt_vertex_fr(S, isMedium, cur, in, sdir, fr, nextPdf, pdf);
    if (isMedium) costheta = 1.f;
    V3 Lr = zero + cur.beta * tr * fr * we * costheta / cameraPdf;
    if (is_black(Lr)) return zero;
    V3 cam_pos = v3(S.cam.pos), cam_n = -v3(S.cam.w);
    float pdfW = pdf_camera_w(S.cam, -sdir);
    float cur_rev = convert_pdf(pdfW, cam_pos, cur.pos, cur.n);
    float prev_rev = convert_pdf(pdf, cur.pos, prev_pos, prev_n);
    (void)cam_n;  // cameraPath[0].rev is written by the reference here and never read by MisWeight
    float mis = bdpt_mis(V, g, s, t, 0.f, 0.f, cur_rev, prev_rev, 0.f);
    return mis * Lr;
  }
  // An inner vertex of each path joined by a shadow ray (:1842-1906).  Here the reference tells a
  // medium vertex from a surface vertex by the vertex's medium, not by its material: a surface
  // reached through a medium is given that medium's phase function.
  BVert c1, l1;
  bv_load(V, g, s - 1, c1);
  bv_load(V, g, LB + t - 1, l1);
  V3 c2_pos = bv_pos(V, g, s - 2), c2_n = v3(V.v[bv_at(V, g, 1, s - 2)]);
  V3 l2_pos = bv_pos(V, g, LB + t - 2), l2_n = v3(V.v[bv_at(V, g, 1, LB + t - 2)]);
  V3 l1Tol2 = normalize(l2_pos - l1.pos);
  V3 l1Toc1 = normalize(c1.pos - l1.pos);
  V3 c1Tol1 = -l1Toc1;
  V3 c1Toc2 = normalize(c2_pos - c1.pos);
  V3 dir = c1.pos - l1.pos;
  bool c1Med = c1.medium != -1, l1Med = l1.medium != -1;
  float stmax = length(dir) - eps;
  if (!c1Med && c1.delta) return zero;
  if (!l1Med && l1.delta) return zero;
  V3 tr = transmittance(S, c1.pos, c1Tol1, eps, stmax, c1.medium, st, stats, n_tr, rng);
  if (is_black(tr)) return zero;
  float cos1 = l1Med ? 1.f : fabsf(dot(l1Toc1, l1.n));
  float cos2 = c1Med ? 1.f : fabsf(dot(c1Tol1, c1.n));
  V3 c1Fr, l1Fr;
  float l1Pdf, c1Pdf, l2Pdf, c2Pdf;
  bdpt_vertex_fr(S, c1Med, c1, c1Toc2, c1Tol1, c1Fr, l1Pdf, c2Pdf);
  bdpt_vertex_fr(S, l1Med, l1, l1Tol2, l1Toc1, l1Fr, c1Pdf, l2Pdf);
  V3 G = tr * cos1 * cos2 / dot(dir, dir);
  V3 Lr = zero + c1.beta * c1Fr * G * l1Fr * l1.beta;
  if (is_black(Lr)) return zero;
  float c1_rev = convert_pdf(c1Pdf, l1.pos, c1.pos, c1.n);
  float l1_rev = convert_pdf(l1Pdf, c1.pos, l1.pos, l1.n);
  float l2_rev = convert_pdf(l2Pdf, l1.pos, l2_pos, l2_n);
  float c2_rev = convert_pdf(c2Pdf, c1.pos, c2_pos, c2_n);
  float mis = bdpt_mis(V, g, s, t, c1_rev, c2_rev, l1_rev, l2_rev, 0.f);
  return mis * Lr;
}

// minstd_rand state after k more draws: x * 48271^k mod (2^31 - 1)
__device__ inline uint32_t rng_skip(uint32_t x, uint32_t k) {
  unsigned long long r = x, a = 48271ull;
  const unsigned long long m = 2147483647ull;
  while (k) {
    if (k & 1u) r = r * a % m;
    a = a * a % m;
    k >>= 1;
  }
  return (uint32_t)r;
}

// Every sample of iterations [first_iter, first_iter + n_iter) whose pixel lies in rows [y0, y1).
// The joins that end at the pinhole (s == 1) land on other pixels and all contributions of all
// streams meet in the film sum: float atomics, as in the reference (:1944-1953).
//
// A wavefront walks the two sub-paths of 64 samples, one per lane, and then shares out the joins
// of all 64 among its lanes: the number of joins of a sample, nCamera x (nLight + 1), varies from
// 2 to several hundred, and a lane that kept to its own sample would idle for most of the time.
// Any lane can evaluate any join (the vertices are columns in HBM) provided it knows where the
// sample's random stream stands at that join: without a density grid only the joins to a sampled
// light point draw numbers, three each, so the stream position is a function of (s, t) and is
// reached with a skip-ahead of the generator.  With a density grid in the scene the transmittance
// of a join draws a data-dependent number of values, and every lane evaluates its own sample's
// joins in order.
__global__ __launch_bounds__(kWave) void k_bdpt(LaunchCtx L, uint32_t first_iter, uint32_t n_iter, int y0, int y1, float* acc,
                                                WalkAux aux, BdptStore V) {
  __shared__ uint2 s_stack[kLdsDepth][kWave];
  __shared__ uint32_t s_end[kWave + 1];  // joins of the samples of lanes < i
  __shared__ uint32_t s_rng[kWave];
  __shared__ int s_nl[kWave], s_nc[kWave];
  __shared__ float s_own[kWave][3];      // what the joins with s >= 2 add to each sample's own pixel
  Stack st{s_stack, aux.spill + (blockIdx.x * kWave + threadIdx.x), aux.spill_stride, 0};
  const DevScene& S = L.S;
  const uint32_t lane = threadIdx.x;
  const uint32_t g0 = blockIdx.x * kWave, g = g0 + lane;
  const int LB = kBdptPathVerts;
  uint32_t n_ext = 0, n_tr = 0;
  uint32_t total = (uint32_t)S.slots_per_iter * n_iter;  // a multiple of the wavefront size: the loop is wave-uniform
  for (uint32_t idx = g; idx < total; idx += gridDim.x * kWave) {
    uint32_t k, local;
    batch_slot_split(S, idx, n_iter, k, local);
    uint32_t px, py;
    const bool live = slot_pixel(S, local, px, py) && (int)py >= y0 && (int)py < y1;
    uint32_t pixel = px + py * (uint32_t)S.width;
    uint32_t rng = 1u;
    int nCamera = 0, nLight = 0;
    if (live) {
      rng = rng_seed(pixel, first_iter + k);
      {  // GenerateCameraPath (:1401-1430): two jitter draws, no lens sample
        float offsetx = rng_uniform(rng) - 0.5f;
        float offsety = rng_uniform(rng) - 0.5f;
        V3 orig, dir;
        primary_ray(S.cam, px + offsetx, py + offsety, 0.f, 0.f, orig, dir);
        BVert x;
        x.pos = v3(S.cam.pos); x.n = -v3(S.cam.w); x.beta = v3(1.f, 1.f, 1.f); x.mat = -1; x.light = -1;
        x.dpdu = v3(0.f, 0.f, 0.f); x.uv = V2{0.f, 0.f}; x.medium = S.cam.medium; x.delta = 0; x.nd = 1; x.fwd = 1.f; x.rev = 0.f;
        bv_store(V, g, 0, x);
        float forward = pdf_camera_w(S.cam, dir);
        nCamera = bdpt_walk(S, st, L.stats, V, g, 0, orig, dir, S.cam.medium, v3(1.f, 1.f, 1.f), forward, true, x.pos, x.n,
                            x.fwd, rng, n_ext);
      }
      {  // GenerateLightPath (:1539-1568)
        float choicePdf;
        int li = pick_light(S, rng_uniform(rng), choicePdf);
        float ux = rng_uniform(rng);
        float uy = rng_uniform(rng);
        float uz = rng_uniform(rng);
        float uw = rng_uniform(rng);
        if (li < S.n_lights) {
          LightRec l;
          load_light(S, li, l);
          int lmed = __float_as_int(S.lights[(size_t)li * kLightF4 + 5].w);
          float su1 = sqrtf(ux);
          V2 uv = V2{1.f - su1, uy * su1};
          V3 pos = uv.x * l.v0 + uv.y * l.v1 + (1 - uv.x - uv.y) * l.v2;
          V3 nor = normalize(uv.x * l.n0 + uv.y * l.n1 + (1 - uv.x - uv.y) * l.n2);
          float pdfW;
          V3 dir = cosine_hemisphere(uz, uw, pdfW);
          V3 uu, ww;
          make_coordinate(nor, uu, ww);
          dir = to_world(dir, uu, ww, nor);
          float pdfA = 1.f / l.area;
          BVert x;
          x.pos = pos; x.n = nor; x.beta = l.radiance; x.mat = -1; x.light = li;
          x.dpdu = v3(0.f, 0.f, 0.f); x.uv = V2{0.f, 0.f}; x.medium = lmed; x.delta = 0; x.nd = 1; x.fwd = pdfA * choicePdf; x.rev = 0.f;
          bv_store(V, g, LB, x);
          V3 beta = v3(1.f, 1.f, 1.f) * (l.radiance * fabsf(dot(dir, nor)) / (pdfA * pdfW * choicePdf));
          nLight = bdpt_walk(S, st, L.stats, V, g, LB, pos, dir, lmed, beta, pdfW, false, pos, nor, x.fwd, rng, n_ext);
        }
      }
    }
    if (S.has_hetero) {
      // joins in the reference's order on the sample's own lane (:1934-1955)
      V3 own = v3(0.f, 0.f, 0.f);
      for (int s = 1; s <= nCamera; ++s) {
        for (int t = 0; t <= nLight; ++t) {
          if (s == 1 && t <= 1) continue;
          int raster = -1;
          V3 Lc = bdpt_connect(S, st, L.stats, V, g, s, t, raster, rng, n_tr);
          if (has_inf(Lc) || has_nan(Lc) || is_black(Lc)) continue;
          if (s == 1) splat(S, acc, raster, Lc);
          else own = own + Lc;
        }
      }
      if (!is_black(own)) splat(S, acc, (int)pixel, own);
      continue;
    }
    // The wavefront's joins are dealt to its lanes round-robin, one kind of join at a time (emitter
    // hits, sampled light points, pinhole, inner vertices: the four branches of Connect()), so that
    // the lanes of a step run the same branch.
    __syncthreads();  // the previous round's readers of the tables below are done
    s_rng[lane] = rng;
    s_nl[lane] = nLight;
    s_nc[lane] = nCamera;
    s_own[lane][0] = 0.f; s_own[lane][1] = 0.f; s_own[lane][2] = 0.f;
    for (int kind = 0; kind < 4; ++kind) {
      {
        const int cs = nCamera - 1 > 0 ? nCamera - 1 : 0, lt = nLight - 1 > 0 ? nLight - 1 : 0;
        uint32_t mine = (uint32_t)(kind == 0 ? cs : kind == 1 ? (nLight >= 1 ? cs : 0) : kind == 2 ? (nCamera >= 1 ? lt : 0) : cs * lt);
        uint32_t incl = mine;
        for (int off = 1; off < kWave; off <<= 1) {
          uint32_t up = (uint32_t)__shfl_up((int)incl, off, kWave);
          if ((int)lane >= off) incl += up;
        }
        __syncthreads();  // the previous kind's readers of s_end are done
        s_end[lane + 1] = incl;
        if (lane == 0) s_end[0] = 0;
        __syncthreads();
      }
      const uint32_t wave_joins = s_end[kWave];
      for (uint32_t j = lane; j < wave_joins; j += kWave) {
        uint32_t lo = 0, hi = kWave - 1;  // the sample i with s_end[i] <= j < s_end[i + 1]
        while (lo < hi) {
          uint32_t mid = (lo + hi + 1) >> 1;
          if (s_end[mid] <= j) lo = mid;
          else hi = mid - 1;
        }
        const uint32_t i = lo;
        const int nl = s_nl[i];
        const uint32_t in_sample = j - s_end[i];
        int s, t;
        if (kind == 0) { s = 2 + (int)in_sample; t = 0; }
        else if (kind == 1) { s = 2 + (int)in_sample; t = 1; }
        else if (kind == 2) { s = 1; t = 2 + (int)in_sample; }
        else { s = 2 + (int)(in_sample / (uint32_t)(nl - 1)); t = 2 + (int)(in_sample % (uint32_t)(nl - 1)); }
        // draws of the joins before this one in the reference's order: three for every (s', 1) with
        // 2 <= s' < s, three more past (s, 1)
        uint32_t draws = nl >= 1 ? 3u * (uint32_t)(s - 2 > 0 ? s - 2 : 0) + (s >= 2 && t > 1 ? 3u : 0u) : 0u;
        uint32_t jr = rng_skip(s_rng[i], draws);
        int raster = -1;
        V3 Lc = bdpt_connect(S, st, L.stats, V, g0 + i, s, t, raster, jr, n_tr);
        if (has_inf(Lc) || has_nan(Lc) || is_black(Lc)) continue;
        if (s == 1) {
          splat(S, acc, raster, Lc);
        } else {  // gathered in LDS: one add to the film per sample instead of one per join
          atomicAdd(&s_own[i][0], Lc.x); atomicAdd(&s_own[i][1], Lc.y); atomicAdd(&s_own[i][2], Lc.z);
        }
      }
    }
    __syncthreads();
    {
      V3 own = v3(s_own[lane][0], s_own[lane][1], s_own[lane][2]);
      if (live && !is_black(own)) splat(S, acc, (int)pixel, own);
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    n_tr += (uint32_t)__shfl_down((int)n_tr, off, 64);
    n_ext += (uint32_t)__shfl_down((int)n_ext, off, 64);
  }
  if ((threadIdx.x & 63u) == 0) {
    if (n_tr) atomicAdd(&L.stats->rays[3], (unsigned long long)n_tr);
    if (n_ext) atomicAdd(&L.stats->rays[0], (unsigned long long)n_ext);
  }
}

// ---- volumetric path tracing on queues ---------------------------------------
// The same integrator with its shadow-ray transmittance walks and light probes
// sent through the traversal kernel instead of being walked lane by lane inside
// the shade kernel (which leaves that kernel at 14 % lane utilisation).  Used
// when no medium of the scene is a density grid: Tr() then draws no random
// numbers, so it can run after the vertex has drawn all of its own.
//
// A vertex leaves behind the OPERANDS of its light-sampled term (the product
// contains the transmittance in the middle, pathtracer.cu:1086 and :1141, and
// must be evaluated in the reference's order once the walk is over), the shadow
// ray with the medium it starts in, and the probe ray.  k_tr_step advances every
// shadow ray by one segment of Tr() (pathtracer.cu:307-331) after each closest-hit
// pass; the next vertex of the path (or k_finish) evaluates the term.

// Li with the pending terms of the previous vertex added, as Volpath() adds them.
__device__ inline V3 resolve_pending_vpt(const PathState& P, int slot, V3 Li, uint32_t flags) {
  float4 px = P.pend_x[slot];
  if (flags & kFPendingM) {
    // Li += tr * beta * phase * radiance / (lightPdf * choicePdf)   (pathtracer.cu:1086)
    float4 pz = P.pend_z[slot];
    V3 tr = v3(P.pend_w[slot]);
    return Li + tr * v3(px) * px.w * v3(pz) / pz.w;
  }
  V3 Ld = v3(0.f, 0.f, 0.f);
  if (flags & kFPendingX) {
    // Ld += weight * tr * fr * radiance * |cos| / (lightPdf * choicePdf)   (pathtracer.cu:1141)
    float4 pz = P.pend_z[slot];
    V3 tr = v3(P.pend_w[slot]);
    float c = P.pend_v[slot].y;
    Ld = Ld + px.w * tr * v3(px) * v3(pz) * c / pz.w;
  }
  V3 y = (flags & kFPendingY) ? v3(P.pend_y[slot]) : v3(0.f, 0.f, 0.f);
  Ld = Ld + y;
  return Li + v3(P.beta_prev[slot]) * Ld;
}

__global__ __launch_bounds__(kWave, CASTER_SHADE_WAVES) void k_shade_vptq(LaunchCtx L, RayQueue cur, ShadeQueues out, StageArea stage) {
  const DevScene& S = L.S;
  const PathState& P = L.P;
  const uint32_t count = *cur.count;
  uint32_t r0, r1;
  wave_rounds(count, r0, r1);
  const uint32_t sbase = blockIdx.x * stage.region_cap;
  uint32_t n_next = 0, n_shadow = 0, n_probe = 0;
  for (uint32_t r = r0; r < r1; ++r) {
    uint32_t idx = r * kWave + threadIdx.x;
    bool want_next = false, want_shadow = false, want_probe = false;
    int slot = 0;
    V3 no = v3(0, 0, 0), nd = v3(0, 0, 1), so = no, sd = nd, po = no, pd = nd;
    float stmax = 0.f;
    if (idx < count) {
      slot = cur.slot[idx];
      float4 o4 = cur.o[idx], d4 = cur.d[idx], h4 = cur.hit[idx];
      if (d4.w >= 0.f || __float_as_int(h4.y) >= 0) {
        V3 ro = v3(o4), rd = v3(d4);
        Hit h;
        h.t = h4.x; h.prim = __float_as_int(h4.y); h.b1 = h4.z; h.b2 = h4.w;
        float4 br = P.beta_rng[slot], lf = P.li_flags[slot];
        V3 beta = v3(br), Li = v3(lf);
        uint32_t rng = __float_as_uint(br.w), flags = __float_as_uint(lf.w);
        int bounces = (int)(flags & kFBounceMask);
        bool specular = (flags & kFSpecular) != 0;
        int rmed = (int)((flags & kFMediumMask) >> kFMediumShift) - 1;
        int crossings = (int)__float_as_uint(P.beta_prev[slot].w);
        if (flags & kFPending) Li = resolve_pending_vpt(P, slot, Li, flags);
        uint32_t pending = 0;
        int nmed = rmed;
        bool alive = true;
        bool count_bounce = true;
        if (h.prim < 0) {
          if ((bounces == 0 || specular) && S.inf.valid) Li = Li + beta * inf_le(S.inf, rd);
          alive = false;
        } else {
          Isect is;
          load_isect(S, ro, rd, h, is);
          V3 pos = is.pos, nor = is.n, dpdu = is.dpdu;
          V2 uv = is.uv;
          float rtmax = h.t;  // the reference's r.tmax after Intersect
          float sampledDist = 0.f;
          bool sampledMedium = false;
          MediumRec med;
          med.g = 0.f;
          if (rmed != -1) {
            // Homogeneous::Sample (medium.h:18-38)
            load_medium(S, rmed, med);
            int channel = fps::f2i_(rng_uniform(rng) * 3);
            channel = channel == 3 ? 2 : channel;
            float sT = channel == 0 ? med.sigmaT.x : (channel == 1 ? med.sigmaT.y : med.sigmaT.z);
            float dist = -fps::log_(1 - rng_uniform(rng)) / sT;
            dist = dist < rtmax ? dist : rtmax;
            sampledMedium = dist < rtmax;
            if (sampledMedium) sampledDist = dist;
            V3 Tr = vexp(med.sigmaT * -dist);
            V3 density = sampledMedium ? (med.sigmaT * Tr) : Tr;
            float pdf = 0;
            pdf += density.x; pdf += density.y; pdf += density.z;
            pdf /= 3.f;
            if (pdf == 0) pdf = 1;
            beta = beta * (sampledMedium ? (Tr * med.sigmaS / pdf) : (Tr / pdf));
          }
          if (is_black(beta)) {
            alive = false;
          } else if (sampledMedium) {
            V3 samplePos = ro + sampledDist * rd;
            V3 radiance, ldir;
            float ltmax, lightPdf, choicePdf;
            sample_one_light(S, rng, samplePos, radiance, ldir, ltmax, lightPdf, choicePdf);
            // Medium::Phase / SamplePhase (medium.h:146-186)
            float phase;
            if (med.g == 0) phase = ONE_OVER_FOUR_PI;
            else {
              float costheta = dot(-rd, ldir);
              float cubicTerm = (1.f + med.g * med.g - 2.f * med.g * costheta);
              phase = ONE_OVER_FOUR_PI * (1.f - med.g * med.g) / sqrtf(cubicTerm * cubicTerm * cubicTerm);
            }
            if (!is_black(radiance)) {
              want_shadow = true;
              so = samplePos; sd = ldir; stmax = ltmax;
              P.pend_x[slot] = f4(beta, phase);
              P.pend_z[slot] = f4(radiance, lightPdf * choicePdf);
              P.pend_w[slot] = f4(v3(1.f, 1.f, 1.f), (uint32_t)rmed);
              pending = kFPending | kFPendingM;
            }
            float ux = rng_uniform(rng);
            float uy = rng_uniform(rng);
            V3 dir;
            if (med.g == 0) {
              float pdf;
              dir = uniform_sphere(ux, uy, pdf);
            } else {
              float g = med.g, costheta;
              if (fabsf(g) < 1e-3f) costheta = 1.f - 2.f * ux;
              else {
                float sqrtTerm = (1.f - g * g) / (1.f - g + 2.f * g * ux);
                costheta = (1.f + g * g - sqrtTerm * sqrtTerm) / (2.f * g);
              }
              float sintheta = sqrtf(1.f - costheta * costheta);
              float phi = TWOPI * uy;
              float sinphi = fps::sin_(phi), cosphi = fps::cos_(phi);
              dir = v3(sintheta * cosphi, sintheta * sinphi, costheta);
            }
            no = samplePos; nd = dir; nmed = rmed;
            specular = false;
          } else {
            bool emitter = (bounces == 0 || specular) && is.light != -1;
            if (emitter) {
              V3 tr = v3(1.f, 1.f, 1.f);
              if (rmed != -1) tr = vexp(med.sigmaT * (-rtmax));
              LightRec l;
              load_light(S, is.light, l);
              V3 le = dot(nor, -rd) > 0.f ? l.radiance : v3(0, 0, 0);
              Li = Li + tr * beta * le;
              alive = false;
            } else if (is.mat == -1) {
              if (++crossings > kMaxCrossings) alive = false;
              count_bounce = false;
              nmed = dot(rd, is.n) > 0 ? is.med_out : is.med_in;
              no = pos; nd = rd;
            } else {
              Mat m;
              load_mat(S, is.mat, m);
              resolve_texture(S, m, uv);
              if (!is_delta(m.type)) {
                V3 radiance, ldir;
                float ltmax, lightPdf, choicePdf;
                sample_one_light(S, rng, pos, radiance, ldir, ltmax, lightPdf, choicePdf);
                float cosl = 0.f;
                if (!is_black(radiance)) {
                  V3 fr;
                  float samplePdf;
                  eval_bsdf(S, m, -rd, ldir, nor, uv, dpdu, fr, samplePdf);
                  float weight = power_heuristic(lightPdf * choicePdf, samplePdf);
                  want_shadow = true;
                  so = pos; sd = ldir; stmax = ltmax;
                  cosl = fabsf(dot(nor, ldir));
                  P.pend_x[slot] = f4(fr, weight);
                  P.pend_z[slot] = f4(radiance, lightPdf * choicePdf);
                  P.pend_w[slot] = f4(v3(1.f, 1.f, 1.f), (uint32_t)rmed);
                  pending |= kFPendingX;
                }
                V3 us;
                us.x = rng_uniform(rng); us.y = rng_uniform(rng); us.z = rng_uniform(rng);
                V3 mo, mfr;
                float mpdf;
                sample_bsdf(S, m, -rd, nor, uv, dpdu, us, mo, mfr, mpdf);
                float absdot = 0.f;
                if (!is_black(mfr) && mpdf != 0) {
                  want_probe = true;
                  po = pos; pd = mo;
                  absdot = fabsf(dot(mo, nor));
                  P.pend_y[slot] = f4(mfr, mpdf);
                  pending |= kFPendingY;
                }
                // Li += beta * Ld: with neither part present that adds beta * 0, which only
                // matters for a non-finite beta (the sample then turns NaN and is dropped)
                bool finite_beta = fabsf(beta.x) < INFINITY && fabsf(beta.y) < INFINITY && fabsf(beta.z) < INFINITY;
                if (pending || !finite_beta) {
                  pending |= kFPending;
                  P.pend_v[slot] = make_float4(absdot, cosl, __int_as_float(rmed), 0.f);
                }
              }
              V3 u;
              u.x = rng_uniform(rng); u.y = rng_uniform(rng); u.z = rng_uniform(rng);
              V3 wo2, fr2;
              float pdf2;
              // the pending term keeps the throughput the vertex was reached with
              V3 beta_here = beta;
              sample_bsdf(S, m, -rd, nor, uv, dpdu, u, wo2, fr2, pdf2);
              if (is_black(fr2)) {
                alive = false;
              } else {
                beta = beta * (fr2 * fabsf(dot(nor, wo2)) / pdf2);
                specular = is_delta(m.type);
                int mm = dot(wo2, nor) > 0 ? is.med_out : is.med_in;
                mm = dot(-rd, nor) * dot(wo2, nor) > 0 ? rmed : mm;
                nmed = mm;
                no = pos; nd = wo2;
              }
              if (pending & kFPending) {
                P.beta_prev[slot] = f4(beta_here, (uint32_t)crossings);
                pending |= 1u << 31;  // beta_prev written
              }
            }
          }
          if (alive && count_bounce) {
            if (bounces > 3) {
              float illumate = fps::clamp_(1.f - luminance(beta), 0.f, 1.f);
              if (rng_uniform(rng) < illumate) alive = false;
              else beta = beta / (1 - illumate);
            }
          }
          if (count_bounce) {
            if (bounces + 1 >= S.maxDepth) alive = false;
            bounces += 1;
          }
        }
        want_next = alive;
        if (!(pending & (1u << 31))) P.beta_prev[slot] = make_float4(0, 0, 0, __uint_as_float((uint32_t)crossings));
        pending &= ~(1u << 31);
        flags = (uint32_t)bounces | (specular ? kFSpecular : 0u) | ((uint32_t)(nmed + 1) << kFMediumShift) | pending;
        P.beta_rng[slot] = f4(beta, rng);
        P.li_flags[slot] = f4(Li, flags);
      }
    }
    stage_push(stage.next, sbase, n_next, want_next, slot, no, -1, nd, INFINITY);
    stage_push(stage.shadow, sbase, n_shadow, want_shadow, slot, so, -1, sd, stmax);
    stage_push(stage.probe, sbase, n_probe, want_probe, slot, po, -1, pd, INFINITY);
  }
  __syncthreads();
  stage_flush(stage.next, sbase, n_next, out.next);
  stage_flush(stage.shadow, sbase, n_shadow, out.shadow);
  stage_flush(stage.probe, sbase, n_probe, out.probe);
}

// One segment of Tr() (pathtracer.cu:307-331) for every shadow ray whose closest
// hit has just been found: an opaque hit ends the ray with zero transmittance,
// otherwise the segment's medium attenuates it and, at an index-matched boundary,
// the ray goes on from the hit in the medium on the other side.
__global__ __launch_bounds__(kWave) void k_tr_step(LaunchCtx L, RayQueue cur, RayQueue next, StageArea stage) {
  const DevScene& S = L.S;
  const PathState& P = L.P;
  const uint32_t count = *cur.count;
  uint32_t r0, r1;
  wave_rounds(count, r0, r1);
  const uint32_t sbase = blockIdx.x * stage.region_cap;
  uint32_t n_next = 0;
  for (uint32_t r = r0; r < r1; ++r) {
    uint32_t idx = r * kWave + threadIdx.x;
    bool want = false;
    int slot = 0;
    V3 o = v3(0, 0, 0), d = v3(0, 0, 1);
    float remaining = 0.f;
    if (idx < count) {
      slot = cur.slot[idx];
      float4 o4 = cur.o[idx], d4 = cur.d[idx], h4 = cur.hit[idx];
      o = v3(o4); d = v3(d4);
      Hit h;
      h.t = h4.x; h.prim = __float_as_int(h4.y); h.b1 = h4.z; h.b2 = h4.w;
      float4 w4 = P.pend_w[slot];
      V3 tr = v3(w4);
      int medium = __float_as_int(w4.w);
      bool found = h.prim >= 0;
      int med_in = -1, med_out = -1;
      V3 n = v3(0, 0, 0);
      bool opaque = false;
      if (found) {
        Isect is;
        load_isect(S, o, d, h, is);
        med_in = is.med_in; med_out = is.med_out; n = is.n;
        opaque = is.mat != -1;
      }
      if (opaque) {
        tr = v3(0, 0, 0);
      } else {
        float ray_tmax = found ? h.t : d4.w;
        if (medium != -1) {
          MediumRec m;
          load_medium(S, medium, m);
          tr = tr * vexp(m.sigmaT * (-ray_tmax));
        }
        if (found) {
          medium = dot(d, n) > 0 ? med_out : med_in;
          remaining = d4.w - ray_tmax;
          o = o + ray_tmax * d;
          want = true;
        }
      }
      P.pend_w[slot] = f4(tr, (uint32_t)medium);
    }
    stage_push(stage.shadow, sbase, n_next, want, slot, o, -1, d, remaining);
  }
  __syncthreads();
  stage_flush(stage.shadow, sbase, n_next, next);
}

// The shadow rays still under way after the queue passes (rays that cross more
// than a few index-matched boundaries are rare) are walked to their end here,
// lane by lane, so that a round of the integrator needs no queue length on the
// host.
__global__ __launch_bounds__(kWave) void k_tr_finish(LaunchCtx L, RayQueue q, WalkAux aux) {
  __shared__ uint2 s_stack[kLdsDepth][kWave];
  Stack st{s_stack, aux.spill + (blockIdx.x * kWave + threadIdx.x), aux.spill_stride, 0};
  const DevScene& S = L.S;
  const PathState& P = L.P;
  const uint32_t count = *q.count;
  uint32_t segs = 0;
  for (uint32_t idx = blockIdx.x * kWave + threadIdx.x; idx < count; idx += gridDim.x * kWave) {
    int slot = q.slot[idx];
    float4 o4 = q.o[idx], d4 = q.d[idx];
    V3 o = v3(o4), d = v3(d4);
    float seg_tmax = d4.w;
    float4 w4 = P.pend_w[slot];
    V3 tr = v3(w4);
    int medium = __float_as_int(w4.w);
    for (int guard = 0; guard < kMaxCrossings; ++guard) {
      Hit h;
      ++segs;
      bool found = traverse<false>(S, o, d, S.epsilon, seg_tmax, h, st, &L.stats->stack_overflow);
      int med_in = -1, med_out = -1;
      V3 n = v3(0, 0, 0);
      if (found) {
        Isect is;
        load_isect(S, o, d, h, is);
        med_in = is.med_in; med_out = is.med_out; n = is.n;
        if (is.mat != -1) { tr = v3(0, 0, 0); break; }
      }
      float ray_tmax = found ? h.t : seg_tmax;
      if (medium != -1) {
        MediumRec m;
        load_medium(S, medium, m);
        tr = tr * vexp(m.sigmaT * (-ray_tmax));
      }
      if (!found) break;
      medium = dot(d, n) > 0 ? med_out : med_in;
      seg_tmax = seg_tmax - ray_tmax;
      o = o + ray_tmax * d;
    }
    P.pend_w[slot] = f4(tr, (uint32_t)medium);
  }
  for (int off = 32; off > 0; off >>= 1) segs += (uint32_t)__shfl_down((int)segs, off, 64);
  if ((threadIdx.x & 63u) == 0 && segs) atomicAdd(&L.stats->rays[3], (unsigned long long)segs);
}

// BSDF-sampled direct light of the volumetric integrator (pathtracer.cu:1148-1194)
// after the probe rays have been traced: as k_probe_resolve, with the
// transmittance of the probe ray's one segment in the vertex's medium.
__global__ __launch_bounds__(kWave) void k_probe_resolve_vpt(LaunchCtx L, RayQueue q) {
  const DevScene& S = L.S;
  const uint32_t count = *q.count;
  for (uint32_t idx = blockIdx.x * kWave + threadIdx.x; idx < count; idx += gridDim.x * kWave) {
    float4 o4 = q.o[idx], d4 = q.d[idx], h4 = q.hit[idx];
    int slot = q.slot[idx];
    V3 pos = v3(o4), out = v3(d4);
    Hit h;
    h.t = h4.x; h.prim = __float_as_int(h4.y); h.b1 = h4.z; h.b2 = h4.w;
    float4 py = L.P.pend_y[slot];
    V3 fr = v3(py);
    float pdf = py.w;
    float4 pv = L.P.pend_v[slot];
    float absdot = pv.x;
    int rmed = __float_as_int(pv.z);
    V3 y = v3(0, 0, 0);
    if (h.prim >= 0) {
      int light = __float_as_int(S.shade[(size_t)h.prim * kShadeF4 + 6].y);
      uint32_t type = __float_as_uint(S.shade[(size_t)h.prim * kShadeF4 + 7].x) & kPrimTypeMask;
      if (type == 0u && light != -1) {
        LightRec l;
        load_light(S, light, l);
        const float4* sh = S.shade + (size_t)h.prim * kShadeF4;
        V3 n = normalize(v3(sh[0]) * (1.f - h.b1 - h.b2) + v3(sh[1]) * h.b1 + v3(sh[2]) * h.b2);
        V3 p = pos + h.t * out;
        V3 radiance = dot(n, -out) > 0.f ? l.radiance : v3(0, 0, 0);
        if (!is_black(radiance)) {
          float pdfA = 1.f / l.area;
          float choicePdf = S.light_cdf[light + 1] - S.light_cdf[light];
          float lenSquare = dot(p - pos, p - pos);
          float costheta = fabsf(dot(n, out));
          float lPdf = pdfA * lenSquare / (costheta);
          float weight = power_heuristic(pdf, lPdf * choicePdf);
          V3 tr = v3(1.f, 1.f, 1.f);
          if (rmed != -1) {
            MediumRec m;
            load_medium(S, rmed, m);
            tr = vexp(m.sigmaT * (-h.t));
          }
          y = weight * tr * fr * radiance * absdot / pdf;
        }
      }
    } else if (S.inf.valid) {
      V3 radiance = inf_le(S.inf, out);
      float choicePdf = S.light_cdf[S.n_lights + 1] - S.light_cdf[S.n_lights];
      float weight = power_heuristic(pdf, ONE_OVER_FOUR_PI * choicePdf);
      V3 tr = v3(1.f, 1.f, 1.f);
      if (rmed != -1) {
        MediumRec m;
        load_medium(S, rmed, m);
        tr = vexp(m.sigmaT * (-INFINITY));
      }
      y = weight * tr * fr * radiance * absdot / pdf;
    }
    L.P.pend_y[slot] = f4(y, 0.f);
  }
}

__global__ void k_add_stat(Stats* st, int which, const uint32_t* n) { st->rays[which] += *n; }

// ----------------------------------------------------------------- film
// Ends every path of the batch and accumulates the film as Output() does
// (pathtracer.cu:2502-2513): acc += colour of iteration first..first+n-1, in
// iteration order.  A NaN/Inf sample adds nothing.  One thread per pixel, so
// the sum order is fixed.
__global__ void k_finish(LaunchCtx L, float* acc, uint32_t n_iter, int y0, int y1) {
  const DevScene& S = L.S;
  uint32_t npix = (uint32_t)S.width * (uint32_t)(y1 - y0);
  uint32_t invalid = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
    uint32_t px = i % (uint32_t)S.width, py = (uint32_t)y0 + i / (uint32_t)S.width;
    uint32_t local = pixel_slot(S, px, py);
    uint32_t pixel = px + py * (uint32_t)S.width;
    V3 a = v3(acc[3 * pixel], acc[3 * pixel + 1], acc[3 * pixel + 2]);
    for (uint32_t k = 0; k < n_iter; ++k) {
      uint32_t slot = batch_slot(S, k, local, n_iter);
      float4 lf = L.P.li_flags[slot];
      V3 Li = v3(lf);
      uint32_t flags = __float_as_uint(lf.w);
      if (flags & kFPending) Li = S.integrator == 2 ? resolve_pending_vpt(L.P, (int)slot, Li, flags) : resolve_pending(L.P, (int)slot, Li, flags);
      bool bad = S.integrator == 0 ? has_nan(Li) : (has_nan(Li) || has_inf(Li));
      if (bad) ++invalid;
      else a = a + Li;
    }
    acc[3 * pixel] = a.x; acc[3 * pixel + 1] = a.y; acc[3 * pixel + 2] = a.z;
  }
  if (invalid) atomicAdd(&L.stats->invalid, (unsigned long long)invalid);
}

// out = tonemap(acc / iter) (pathtracer.cu:196-213, 2512-2516); mode 0 keeps
// the linear mean.
__global__ void k_resolve_film(const float* acc, float* out, uint32_t n_pix, float iter, int mode, int filmic) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pix; i += gridDim.x * blockDim.x) {
    V3 c = v3(acc[3 * i], acc[3 * i + 1], acc[3 * i + 2]) / iter;
    if (mode == 1) {
      if (filmic) {
        c = c - v3(0.004f, 0.004f, 0.004f);
        c = v3(fps::fmax_(0.f, c.x), fps::fmax_(0.f, c.y), fps::fmax_(0.f, c.z));
        c = (c * (6.2f * c + 0.5f)) / (c * (6.2f * c + 1.7f) + 0.06f);
      } else {
        float g = 1.f / 2.2f, e = 1.41421356f;
        c = v3(fps::fmax_(c.x, 1e-5f), fps::fmax_(c.y, 1e-5f), fps::fmax_(c.z, 1e-5f));
        c = v3(fps::pow_(c.x * e, g), fps::pow_(c.y * e, g), fps::pow_(c.z * e, g));
      }
    }
    out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
  }
}

// Primary-visibility map: primitive id per pixel from the traced camera queue.
__global__ void k_gather_ids(LaunchCtx L, RayQueue q, int32_t* ids) {
  const DevScene& S = L.S;
  uint32_t npix = (uint32_t)S.width * (uint32_t)S.height;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
    uint32_t px = i % (uint32_t)S.width, py = i / (uint32_t)S.width;
    ids[i] = __float_as_int(q.hit[pixel_slot(S, px, py)].y);
  }
}

// Adds the three queue lengths of a finished bounce to the ray statistics and
// clears the queues that the next shade pass appends to.
__global__ void k_roll(Stats* st, const uint32_t* n_ext, const uint32_t* n_shadow, const uint32_t* n_probe,
                       uint32_t* clear0, uint32_t* clear1, uint32_t* clear2, uint32_t ext_override, uint32_t* work,
                       uint32_t n_work, uint32_t work_init) {
  // work counters start at zero (k_trace adds the size of its static assignment itself)
  for (uint32_t i = 0; i < n_work; ++i) work[i * kWorkPad] = work_init;
  if (ext_override) st->rays[0] += ext_override;
  else if (n_ext) st->rays[0] += *n_ext;
  if (n_shadow) st->rays[1] += *n_shadow;
  if (n_probe) st->rays[2] += *n_probe;
  if (clear0) *clear0 = 0;
  if (clear1) *clear1 = 0;
  if (clear2) *clear2 = 0;
}

}  // namespace caster
