// Part of the gfx950 wavefront renderer (included by ../kernels.hip, in this order: common.h, traverse.h, trace.h, shading.h, bsdf.h, path.h, volpath.h, lighttrace.h, bdpt.h, volpath_queues.h, film.h).
// Ray/box and ray/primitive tests, the traversal stack, one-lane closest-hit / any-hit walk.
#pragma once

namespace caster {

// --------------------------------------------------------------- traversal
// What the slab test needs of a ray: origin, 1 / direction, noi = -(origin * inv), and whether the test is made
// with one fused multiply-add per plane (fps::slab_plane_: every ray whose noi is finite) or in the reference's
// own form.
struct SlabRay {
  V3 o, inv, noi;
  bool fused;
};
__device__ inline SlabRay slab_ray(V3 o, V3 d) {
  SlabRay r;
  r.o = o;
  r.inv = v3(1.f / d.x, 1.f / d.y, 1.f / d.z);
  r.noi = v3(-(o.x * r.inv.x), -(o.y * r.inv.y), -(o.z * r.inv.z));
  r.fused = fps::slab_fused_(r.noi.x, r.noi.y, r.noi.z);
  return r;
}
// Slab test of the reference (bbox.h:75-98): tFar widened by 1+2*gamma(3),
// rejected when behind the ray, empty, or beyond the current tmax; tmin is
// not consulted.  Returns tNear through `tn`.
__device__ inline bool box_test(V3 lo, V3 hi, const SlabRay& r, float tmax, float scale, float& tn) {
  // Both forms of fps::slab_plane_ through one sequence, fma(plane - sub, inv, add): the fused form with sub = 0 and
  // add = noi (plane - 0 is plane), the reference's form with sub = origin and add = 0 (a product plus zero is the
  // rounded product, up to the sign of a zero, which no comparison below sees).  sub and add do not change along a
  // walk, so a lane that walks the tree in a loop of its own sets them up once.
  const V3 sub = r.fused ? v3(0.f, 0.f, 0.f) : r.o, add = r.fused ? r.noi : v3(0.f, 0.f, 0.f);
  float txNear = fps::fma_(lo.x - sub.x, r.inv.x, add.x);
  float txFar = fps::fma_(hi.x - sub.x, r.inv.x, add.x);
  float tyNear = fps::fma_(lo.y - sub.y, r.inv.y, add.y);
  float tyFar = fps::fma_(hi.y - sub.y, r.inv.y, add.y);
  float tzNear = fps::fma_(lo.z - sub.z, r.inv.z, add.z);
  float tzFar = fps::fma_(hi.z - sub.z, r.inv.z, add.z);
  // v_min_f32 / v_max_f32 drop a NaN operand exactly as fps::fmin_/fmax_ do;
  // they may differ in the sign of a zero result, which no comparison below sees.
  float tNear = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(txNear, txFar), __builtin_fminf(tyNear, tyFar)),
                                __builtin_fminf(tzNear, tzFar));
  float tFar = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(txNear, txFar), __builtin_fmaxf(tyNear, tyFar)),
                               __builtin_fmaxf(tzNear, tzFar));
  tFar *= scale;
  tn = tNear;
  return !(tFar <= 0.00001f) && !(tNear > tFar) && !(tNear > tmax * fps::kNearCullSlack);
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
// (`sr` holds the ray's origin and what the slab test wants, which the walkers hold anyway: the sphere
// test consults the sphere's own box)
__device__ inline bool prim_test_rec(const DevScene& S, float4 a, float4 b, float4 c, const SlabRay& sr, V3 d, float maxD,
                                     float tmin, float tmax, float& t, float& ob1, float& ob2, bool& last);
__device__ inline bool prim_test(const DevScene& S, int i, const SlabRay& sr, V3 d, float maxD, float tmin, float tmax,
                                 float& t, float& ob1, float& ob2, bool& last) {
  const float4* rec = S.isect + (size_t)i * kIsectF4;
  return prim_test_rec(S, rec[0], rec[1], rec[2], sr, d, maxD, tmin, tmax, t, ob1, ob2, last);
}
__device__ inline bool prim_test(const DevScene& S, int i, V3 o, V3 d, float maxD, float tmin, float tmax,
                                 float& t, float& ob1, float& ob2, bool& last) {
  return prim_test(S, i, slab_ray(o, d), d, maxD, tmin, tmax, t, ob1, ob2, last);
}
__device__ inline bool prim_test_rec(const DevScene& S, float4 a, float4 b, float4 c, V3 o, V3 d, float maxD, float tmin,
                                     float tmax, float& t, float& ob1, float& ob2, bool& last) {
  return prim_test_rec(S, a, b, c, slab_ray(o, d), d, maxD, tmin, tmax, t, ob1, ob2, last);
}
// The same test on a record already in registers.
__device__ inline bool prim_test_rec(const DevScene& S, float4 a, float4 b, float4 c, const SlabRay& sr, V3 d, float maxD,
                                     float tmin, float tmax, float& t, float& ob1, float& ob2, bool& last) {
  const V3 o = sr.o;
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
    float maxS = maxabs3(s);
    float maxS1 = maxabs3(s1);
    float deltaS2 = 2 * (S.g2 * maxS * maxS1 + deltaS * maxS1 + deltaS1 * maxS);
    float maxS2 = maxabs3(s2);
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
    // The roots are distances only for a unit direction, and rays that leave a line or a sphere
    // carry others: a candidate in front of the point where the ray enters the sphere's own
    // bounding box (c -+ r, the box the tree encloses) is no hit.  What remains lies behind the
    // entry of every box above it, so it is tested whenever it could still win, whatever the
    // order the tree is walked in.
    float tn;
    (void)box_test(v3(a) - radius, v3(a) + radius, sr, INFINITY, S.box_scale, tn);
    if (tn > th * fps::kNearCullSlack) return false;
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
  int ctx = 0;   // k_trace keeps two rays per lane: which of the lane's two stacks this is (rows ctx * kLdsDepth of
                 // `lds`, rows ctx * kSpillDepth of the spill column)
#ifdef CASTER_WALK_STATS
  uint32_t n_pop_iters = 0, n_pop_calls = 0, n_spill_push = 0, n_spill_pop = 0, n_slow = 0;   // tuning aid, per lane
#endif
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
  *(lds_u64_t*)&s.lds[s.ctx * kLdsDepth + sp][lane] = ((unsigned long long)e.y << 32) | e.x;
}
__device__ inline uint2 lds_get(Stack& s, int sp, uint32_t lane) {
  unsigned long long v = *(lds_u64_t*)&s.lds[s.ctx * kLdsDepth + sp][lane];
  return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}
#else
__device__ inline void lds_put(Stack& s, int sp, uint32_t lane, uint2 e) { s.lds[s.ctx * kLdsDepth + sp][lane] = e; }
__device__ inline uint2 lds_get(Stack& s, int sp, uint32_t lane) { return s.lds[s.ctx * kLdsDepth + sp][lane]; }
#endif
__device__ inline bool stack_push(Stack& s, uint32_t lane, int ref, float tn) {
  uint2 e = make_uint2((uint32_t)ref, __float_as_uint(tn));
  if (s.sp < kLdsDepth) lds_put(s, s.sp, lane, e);
  else if (s.sp < kLdsDepth + kSpillDepth) {
    s.spill[(size_t)(s.ctx * kSpillDepth + s.sp - kLdsDepth) * s.stride] = e;
#ifdef CASTER_WALK_STATS
    ++s.n_spill_push;
#endif
  } else return false;
  ++s.sp;
  return true;
}
// Next deferred child that can still hold a closer hit: an entry whose entry
// distance exceeds the current tmax is dropped, which is the test the
// reference makes when it reaches that node (pathtracer.cu:231-232,
// bbox.h:95).  kRefEmpty when nothing is left.
__device__ inline int stack_next(Stack& s, uint32_t lane, float tmax) {
#ifdef CASTER_WALK_STATS
  ++s.n_pop_calls;
#endif
  while (s.sp > 0) {
    --s.sp;
    uint2 e;
#ifdef CASTER_WALK_STATS
    ++s.n_pop_iters;
    if (s.sp >= kLdsDepth) ++s.n_spill_pop;
#endif
    if (s.sp < kLdsDepth) e = lds_get(s, s.sp, lane);
    else e = s.spill[(size_t)(s.ctx * kSpillDepth + s.sp - kLdsDepth) * s.stride];
    if (!(__uint_as_float(e.y) > tmax * fps::kNearCullSlack)) return (int)e.x;  // the slab test's own cull, with the ray's present tmax
  }
  return kRefEmpty;
}

// Occlusion rays keep no entry distances (their tmax never shrinks: a box that passed the test once would pass it
// again), so their stack entries are bare references: 4 bytes, and the same LDS holds twice as many of them before
// the column in HBM is needed.  Rows of 64 x 4 bytes, one conflict-free ds_write_b32 / ds_read_b32 per access.
constexpr int kLdsDepthRefs = 2 * kLdsDepth;
typedef __attribute__((address_space(3))) volatile uint32_t lds_u32_t;
__device__ inline lds_u32_t* lds_ref_slot(Stack& s, int sp, uint32_t lane) {
  return (lds_u32_t*)&s.lds[s.ctx * kLdsDepth][0] + (sp * kWave + (int)lane);
}
__device__ inline bool stack_push_ref(Stack& s, uint32_t lane, int ref) {
  if (s.sp < kLdsDepthRefs) *lds_ref_slot(s, s.sp, lane) = (uint32_t)ref;
  else if (s.sp < kLdsDepthRefs + kSpillDepth) s.spill[(size_t)(s.ctx * kSpillDepth + s.sp - kLdsDepthRefs) * s.stride] = make_uint2((uint32_t)ref, 0u);
  else return false;
  ++s.sp;
  return true;
}
__device__ inline int stack_pop_ref(Stack& s, uint32_t lane) {
  if (s.sp <= 0) return kRefEmpty;
  --s.sp;
  if (s.sp < kLdsDepthRefs) return (int)*lds_ref_slot(s, s.sp, lane);
  return (int)s.spill[(size_t)(s.ctx * kSpillDepth + s.sp - kLdsDepthRefs) * s.stride].x;
}
// The next deferred child of a ray whose stack holds (reference, entry distance) pairs (KEYS) or bare references.
template <bool KEYS>
__device__ inline int stack_next_of(Stack& s, uint32_t lane, float tmax) {
  return KEYS ? stack_next(s, lane, tmax) : stack_pop_ref(s, lane);
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
#ifndef CASTER_PARTIAL_SORT
#define CASTER_PARTIAL_SORT 0   // 1: nearest child by a tournament, the losers deferred unsorted (18 instructions less per node
                                // step, 2.7 % more node visits: launches 2 % longer); 0: all four sorted
#endif
#ifndef CASTER_BRANCHY_NODE
#define CASTER_BRANCHY_NODE 0   // 1: the exchange network and the pushes as branches (what the compiler makes of `if`)
#endif
// A ray's direction octant as byte offsets into a node record: of the three float4s that hold the planes its
// direction reaches first (lo.x / hi.x at 0 / 48, lo.y / hi.y at 16 / 64, lo.z / hi.z at 32 / 80) and of the three it
// reaches last, a byte each, packed into two registers.  The traversal kernel adds them to a node's offset with
// byte-selecting adds (one instruction per load address, no unpacking).
struct Octant {
  uint32_t a;   // near x | near y << 8 | near z << 16 | far x << 24
  uint32_t b;   // far y | far z << 8 | kOctPlain when the ray keeps to the plain slab test
};
constexpr uint32_t kOctPlain = 1u << 16;
__device__ inline Octant octant_of(V3 inv, bool plain) {
  const uint32_t nx = inv.x < 0.f ? 48u : 0u, ny = inv.y < 0.f ? 64u : 16u, nz = inv.z < 0.f ? 80u : 32u;
  return Octant{nx | (ny << 8) | (nz << 16) | ((48u - nx) << 24), (80u - ny) | ((112u - nz) << 8) | (plain ? kOctPlain : 0u)};
}
// base + byte K of `packed`
template <int K>
__device__ inline uint32_t add_byte(uint32_t base, uint32_t packed) {
  uint32_t r;
  if (K == 0) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(base), "v"(packed));
  if (K == 1) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(base), "v"(packed));
  if (K == 2) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(base), "v"(packed));
  if (K == 3) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(base), "v"(packed));
  return r;
}
// Such a ray's slab test when its test is the fused one (finite 1 / direction, origin and their product): the
// nearer of an axis's two planes is then known from the sign alone -- for lo <= hi, fma(lo, inv, noi) <=
// fma(hi, inv, noi) when inv > 0 and >= when inv < 0, rounding included, since a single rounding of a monotone
// function is monotone -- so the six minima and maxima of box_test() fall away and the results are the same bit
// for bit.  (An empty child slot, lo = +inf and hi = -inf, gives tNear = +inf and tFar = -inf here and fails
// without a look at its reference.)  With an infinite component, (plane - o) * inv is not a number where the
// origin lies in the plane, box_test()'s minima and maxima drop that operand, and which one they drop decides:
// such rays keep to box_test() and the reference's form of the test.
__device__ inline bool box_test_octant(float nx, float ny, float nz, float fx, float fy, float fz, V3 inv, V3 noi,
                                       float tmax_slack, float scale, float& tn) {
  using fps::fma_;
  float tNear = __builtin_fmaxf(__builtin_fmaxf(fma_(nx, inv.x, noi.x), fma_(ny, inv.y, noi.y)), fma_(nz, inv.z, noi.z));
  float tFar = __builtin_fminf(__builtin_fminf(fma_(fx, inv.x, noi.x), fma_(fy, inv.y, noi.y)), fma_(fz, inv.z, noi.z));
  tFar *= scale;
  tn = tNear;
  return !(tFar <= 0.00001f) && !(tNear > tFar) && !(tNear > tmax_slack);
}
__device__ inline float4 node_f4(const DevScene& S, uint32_t byte_offset) {
  // (uniform base + 32-bit offset: one address instruction per load)
  return *(const float4*)((const char*)S.nodes + byte_offset);
}
// OCT: the lane's ray has finite reciprocal direction and origin, `oct` is its octant (k_trace sees to both).
template <bool SORT = true, bool OCT = false>
__device__ inline int node_step(const DevScene& S, int cur, const SlabRay& sr, float tmax, Stack& st, uint32_t lane,
                                unsigned long long* overflow, const float4* top = nullptr, int n_top = 0,
                                Octant oct = Octant{0u, 0u}) {
  float4 rf;
  float k0, k1, k2, k3;
  bool h0, h1, h2, h3;
  int r0, r1, r2, r3;
  if (OCT) {
    float4 ax, ay, az, bx, by, bz;   // the planes reached first and last, of the four children
    if (cur < n_top) {
      const uint32_t b = (uint32_t)(size_t)top + (uint32_t)cur * (uint32_t)(kTopF4 * 16);   // (an LDS address is 32 bits)
      typedef __attribute__((address_space(3))) const float4 lds_cf4;
      auto at = [](uint32_t a) { return lds_f4((const float4*)(lds_cf4*)(size_t)a); };
      ax = at(add_byte<0>(b, oct.a)); ay = at(add_byte<1>(b, oct.a)); az = at(add_byte<2>(b, oct.a));
      bx = at(add_byte<3>(b, oct.a)); by = at(add_byte<0>(b, oct.b)); bz = at(add_byte<1>(b, oct.b));
      rf = at(b + 96u);
    } else {
      const uint32_t b = (uint32_t)cur * (uint32_t)(kNodeF4 * 16);
      ax = node_f4(S, add_byte<0>(b, oct.a)); ay = node_f4(S, add_byte<1>(b, oct.a)); az = node_f4(S, add_byte<2>(b, oct.a));
      bx = node_f4(S, add_byte<3>(b, oct.a)); by = node_f4(S, add_byte<0>(b, oct.b)); bz = node_f4(S, add_byte<1>(b, oct.b));
      rf = node_f4(S, b + 96u);
    }
    const float ts = tmax * fps::kNearCullSlack;
    h0 = box_test_octant(ax.x, ay.x, az.x, bx.x, by.x, bz.x, sr.inv, sr.noi, ts, S.box_scale, k0);
    h1 = box_test_octant(ax.y, ay.y, az.y, bx.y, by.y, bz.y, sr.inv, sr.noi, ts, S.box_scale, k1);
    h2 = box_test_octant(ax.z, ay.z, az.z, bx.z, by.z, bz.z, sr.inv, sr.noi, ts, S.box_scale, k2);
    h3 = box_test_octant(ax.w, ay.w, az.w, bx.w, by.w, bz.w, sr.inv, sr.noi, ts, S.box_scale, k3);
    r0 = __float_as_int(rf.x); r1 = __float_as_int(rf.y); r2 = __float_as_int(rf.z); r3 = __float_as_int(rf.w);
  } else {
    float4 lx, ly, lz, hx, hy, hz;
    if (cur < n_top) {
      const float4* n = top + cur * kTopF4;
      lx = lds_f4(n); ly = lds_f4(n + 1); lz = lds_f4(n + 2); hx = lds_f4(n + 3); hy = lds_f4(n + 4); hz = lds_f4(n + 5);
      rf = lds_f4(n + 6);
    } else {
      const float4* n = S.nodes + (size_t)cur * kNodeF4;
      lx = n[0]; ly = n[1]; lz = n[2]; hx = n[3]; hy = n[4]; hz = n[5]; rf = n[6];
    }
    r0 = __float_as_int(rf.x); r1 = __float_as_int(rf.y); r2 = __float_as_int(rf.z); r3 = __float_as_int(rf.w);
    h0 = box_test(v3(lx.x, ly.x, lz.x), v3(hx.x, hy.x, hz.x), sr, tmax, S.box_scale, k0) && r0 != kRefEmpty;
    h1 = box_test(v3(lx.y, ly.y, lz.y), v3(hx.y, hy.y, hz.y), sr, tmax, S.box_scale, k1) && r1 != kRefEmpty;
    h2 = box_test(v3(lx.z, ly.z, lz.z), v3(hx.z, hy.z, hz.z), sr, tmax, S.box_scale, k2) && r2 != kRefEmpty;
    h3 = box_test(v3(lx.w, ly.w, lz.w), v3(hx.w, hy.w, hz.w), sr, tmax, S.box_scale, k3) && r3 != kRefEmpty;
  }
  if (!(h0 || h1 || h2 || h3)) return stack_next_of<SORT>(st, lane, tmax);
  [[maybe_unused]] const int nh = (int)h0 + (int)h1 + (int)h2 + (int)h3;
#if CASTER_BRANCHY_NODE
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
#else
  if (!SORT) {
    // An occlusion ray: its tmax never shrinks, so the order in which boxes are visited decides nothing and a
    // deferred box that passed the test once would pass it again: no entry distances, no sorting, and the stack
    // holds bare references.
    // The first child that is hit is visited now, the other hit children are deferred: three stores, each at the
    // height the ones before it left (a store whose child was not hit is overwritten by the next one, or lies
    // beyond the top of the stack and is never read).
    const int first = h0 ? r0 : (h1 ? r1 : (h2 ? r2 : r3));
    const bool d1 = h1 && h0, d2 = h2 && (h0 || h1), d3 = h3 && (h0 || h1 || h2);
    if (st.sp + 3 <= kLdsDepthRefs) {
      int sp = st.sp;
      *lds_ref_slot(st, sp, lane) = (uint32_t)r1; sp += d1 ? 1 : 0;
      *lds_ref_slot(st, sp, lane) = (uint32_t)r2; sp += d2 ? 1 : 0;
      *lds_ref_slot(st, sp, lane) = (uint32_t)r3; sp += d3 ? 1 : 0;
      st.sp = sp;
    } else {
      bool ok = true;
      if (d1) ok &= stack_push_ref(st, lane, r1);
      if (d2) ok &= stack_push_ref(st, lane, r2);
      if (d3) ok &= stack_push_ref(st, lane, r3);
      if (!ok) atomicAdd(overflow, 1ull);
    }
    return first;
  }
#if CASTER_PARTIAL_SORT
  // The nearest hit is found by a tournament (three compares); the three losers are deferred in the order the
  // tournament leaves them -- the loser of each pair below the loser of the final -- and stored at running heights
  // like an occlusion ray's.  What lies on the stack is then not fully sorted: a ray may look at a farther box
  // before a nearer one (never wrongly: entries are still culled against the ray's present tmax when popped).
  k0 = h0 ? k0 : INFINITY; k1 = h1 ? k1 : INFINITY; k2 = h2 ? k2 : INFINITY; k3 = h3 ? k3 : INFINITY;
  const bool c01 = k1 < k0, c23 = k3 < k2;
  const float ka = c01 ? k1 : k0, kb = c01 ? k0 : k1, kc = c23 ? k3 : k2, kd = c23 ? k2 : k3;
  const int ra = c01 ? r1 : r0, rb = c01 ? r0 : r1, rc = c23 ? r3 : r2, rd = c23 ? r2 : r3;
  const bool cf = kc < ka;
  const int nearest = cf ? rc : ra;
  const float kl = cf ? ka : kc;
  const int rl = cf ? ra : rc;
  const bool db = h0 && h1, dd = h2 && h3, dl = (h0 || h1) && (h2 || h3);
  if (st.sp + 3 <= kLdsDepth) {
    int sp = st.sp;
    lds_put(st, sp, lane, make_uint2((uint32_t)rb, __float_as_uint(kb))); sp += db ? 1 : 0;
    lds_put(st, sp, lane, make_uint2((uint32_t)rd, __float_as_uint(kd))); sp += dd ? 1 : 0;
    lds_put(st, sp, lane, make_uint2((uint32_t)rl, __float_as_uint(kl))); sp += dl ? 1 : 0;
    st.sp = sp;
  } else {
#ifdef CASTER_WALK_STATS
    ++st.n_slow;
#endif
    bool ok = true;
    if (db) ok &= stack_push(st, lane, rb, kb);
    if (dd) ok &= stack_push(st, lane, rd, kd);
    if (dl) ok &= stack_push(st, lane, rl, kl);
    if (!ok) atomicAdd(overflow, 1ull);
  }
  return nearest;
#else
  // The children that are hit, farthest first, the misses behind them (their key is -infinity): a
  // five-exchange network written as selects, so that it costs five compares and twenty moves and
  // no branch.  (A key that is not a number never changes places, and neither does an empty child
  // slot, which is always behind the children there are: pairs stay pairs.)  The nearest hit is
  // then item nh-1 and is returned; items 0 .. nh-2 are the deferred ones in the order they must
  // lie on the stack, farthest at the bottom.  Any-hit rays go the same way: what they find first
  // does not matter, near boxes first is as good an order as any.
  k0 = h0 ? k0 : -INFINITY; k1 = h1 ? k1 : -INFINITY; k2 = h2 ? k2 : -INFINITY; k3 = h3 ? k3 : -INFINITY;
#define CASTER_CSWAP(ka, ra, kb, rb)                                   \
  {                                                                     \
    const bool c_ = ka < kb;                                            \
    const float nka_ = c_ ? kb : ka, nkb_ = c_ ? ka : kb;               \
    const int nra_ = c_ ? rb : ra, nrb_ = c_ ? ra : rb;                 \
    ka = nka_; kb = nkb_; ra = nra_; rb = nrb_;                         \
  }
  CASTER_CSWAP(k0, r0, k1, r1)
  CASTER_CSWAP(k2, r2, k3, r3)
  CASTER_CSWAP(k0, r0, k2, r2)
  CASTER_CSWAP(k1, r1, k3, r3)
  CASTER_CSWAP(k1, r1, k2, r2)
#undef CASTER_CSWAP
  const int nearest = nh > 3 ? r3 : (nh > 2 ? r2 : (nh > 1 ? r1 : r0));
  if (st.sp + 3 <= kLdsDepth) {
    // three stores at fixed distances from one address; those beyond the new top of the stack (a
    // miss, or the nearest hit itself) are never read
    lds_put(st, st.sp, lane, make_uint2((uint32_t)r0, __float_as_uint(k0)));
    lds_put(st, st.sp + 1, lane, make_uint2((uint32_t)r1, __float_as_uint(k1)));
    lds_put(st, st.sp + 2, lane, make_uint2((uint32_t)r2, __float_as_uint(k2)));
    st.sp += nh - 1;
  } else {
#ifdef CASTER_WALK_STATS
    ++st.n_slow;
#endif
    bool ok = true;
    if (nh > 1) ok &= stack_push(st, lane, r0, k0);
    if (nh > 2) ok &= stack_push(st, lane, r1, k1);
    if (nh > 3) ok &= stack_push(st, lane, r2, k2);
    if (!ok) atomicAdd(overflow, 1ull);
  }
  return nearest;
#endif
#endif
}

// Tests the primitives of the leaf starting at `i`.  Among hits at equal
// distance the higher primitive index wins, as in the reference's in-order
// scan with its `t <= tmax` acceptance.  ANY: returns at the first hit.
template <bool ANY>
__device__ inline bool leaf_step(const DevScene& S, int i, const SlabRay& sr, V3 d, float maxD, float tmin, Hit& hit) {
  bool last;
  do {
    float t, b1, b2;
    if (prim_test(S, i, sr, d, maxD, tmin, hit.t, t, b1, b2, last)) {
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
  const SlabRay sr = slab_ray(o, d);
  float maxD = maxabs3(d);
  hit.prim = -1;
  hit.t = tmax < S.far_t ? tmax : S.far_t;   // (no hit is looked for beyond the scene's far distance)
  hit.b1 = hit.b2 = 0.f;
  st.sp = 0;
  int cur = S.n_nodes ? 0 : kRefEmpty;
  while (cur != kRefEmpty) {
    if (cur >= 0) {
      cur = node_step(S, cur, sr, hit.t, st, lane, overflow);
    } else {
      if (leaf_step<ANY>(S, ~cur, sr, d, maxD, tmin, hit)) return true;
      cur = stack_next(st, lane, hit.t);
    }
  }
  return hit.prim >= 0;
}

}  // namespace caster
