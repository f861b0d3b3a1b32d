// TEST INFRASTRUCTURE -- not part of the product.
//
// CPU oracle: a plain, scalar restatement of the reference's AO / path /
// volumetric-path / light-tracing / bidirectional integrators and every device function they call.  One
// function per reference function, each citing the lines it follows
// (paths relative to CUDA-PathTracer/tracer/).  The reference does not compile
// and ships no tests, scenes or golden vectors, so parity against it is
// UNPINNED: this file is written from the reference text, pinned by Thrust's
// published minstd_rand known answer and by fixtures it generates itself.
//
// Conventions fixed where the reference leaves them open (each is mirrored
// by the HIP renderer):
//   * every `uniform(rng)` inside one expression is drawn left to right;
//   * transcendental functions and rsqrt come from fpspec.h, fmin/fmax drop
//     NaNs; dot and cross products are fused as nvcc's default contraction
//     fuses them and the slab test is one fused multiply-add per plane
//     (fpspec.h), nothing else is fused;
//   * the Triangle shading frame is n = ns, dpdu = ss (Triangle.h:122-136 does
//     not compile; cross(ss,ts) is parallel to ns so Faceforward returns ns);
//   * a sphere hit whose chosen root lies beyond ray.tmax is rejected (the
//     reference accepts the far root unconditionally, sphere.h:61-66, which
//     would make the result depend on traversal order);
//   * a sphere candidate that lies before the point where the ray enters the
//     sphere's own bounding box is rejected.  sphere.h:24-66 assumes a unit
//     direction; rays leaving a Line primitive (n = -dir, line.h:74-76) or a
//     sphere (dpdu not orthogonal to n, sphere.h:74-80) carry directions of
//     other lengths, the roots are then no distances, and a "hit" reported in
//     front of the box is found or not depending on whether another hit has
//     shortened the ray before the sphere's leaf is reached.  With the rule a
//     candidate at t implies that every box above it is entered at or before
//     t, so it is tested whenever it could still win: the closest hit is a
//     function of the ray alone;
//   * Line hits leave the medium indices at -1 (line.h:74-84 leaves them
//     uninitialised);
//   * a non-finite sample contributes nothing and an AO miss is (1,1,1) (the
//     reference re-accumulates a stale buffer, pathtracer.cu:845,867,1009);
//   * light selection that matches no bin returns the last bin
//     (pathtracer.cu:181-190 falls off the end); a scene without lights draws
//     the light-sampling numbers and gets zero radiance;
//   * at most kMaxCrossings index-matched boundary crossings per path and
//     per transmittance ray;
//   * no hit is looked for farther along a ray than 16384 diagonals of the scene's
//     bounds (Scene::far_t; the reference looks to +infinity, ray.h:12): what the
//     triangle test reports out there is a rounding artefact, and a path that
//     went on from such a point would send rays that single precision cannot
//     tell apart from one another at the scene, each of them visiting the whole
//     tree;
//   * the slab test drops a box entered beyond tmax * (1 + 2^-16), not beyond
//     tmax itself (bbox.h:96): of two triangles sharing the edge a ray hits
//     both are then always tested and the closest hit (smallest t, then
//     highest index) does not depend on the order the tree is walked in.
#include "oracle.h"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../cuda-pathtracer_amd/csrc/scene/scene.h"

namespace {

using caster::Scene;
using caster::Prim;
using caster::Node;
using caster::Material;

thread_local std::string g_err;

// ---------------------------------------------------------------- vectors
struct V3 { float x, y, z; };
struct V2 { float x, y; };
struct V4 { float x, y, z, w; };
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 v3(caster::f3 f) { return V3{f.x, f.y, f.z}; }
inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(V3 a, V3 b) { return V3{a.x / b.x, a.y / b.y, a.z / b.z}; }
inline V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
inline V3 operator+(V3 a, float s) { return V3{a.x + s, a.y + s, a.z + s}; }
inline V3 operator-(V3 a, float s) { return V3{a.x - s, a.y - s, a.z - s}; }
inline V3 operator-(float s, V3 a) { return V3{s - a.x, s - a.y, s - a.z}; }
inline V3& operator+=(V3& a, V3 b) { a = a + b; return a; }
inline V3& operator*=(V3& a, V3 b) { a = a * b; return a; }
inline V3& operator*=(V3& a, float s) { a = a * s; return a; }
inline V3& operator/=(V3& a, float s) { a = a / s; return a; }
inline V4 operator*(float s, V4 a) { return V4{s * a.x, s * a.y, s * a.z, s * a.w}; }
inline V4 operator+(V4 a, V4 b) { return V4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
// cutil_math.h:1136 and :1308, with the fused multiply-adds nvcc's default contraction puts there (fpspec.h)
inline float dot(V3 a, V3 b) { return fps::dot3_(a.x, a.y, a.z, b.x, b.y, b.z); }
inline V3 cross(V3 a, V3 b) {
  return V3{fps::diffprod_(a.y, b.z, a.z, b.y), fps::diffprod_(a.z, b.x, a.x, b.z), fps::diffprod_(a.x, b.y, a.y, b.x)};
}
inline float length(V3 v) { return sqrtf(dot(v, v)); }                       // cutil_math.h:1179
inline V3 normalize(V3 v) { return v * fps::rsqrt_(dot(v, v)); }             // cutil_math.h:1197
inline V3 vabs(V3 v) { return V3{fabsf(v.x), fabsf(v.y), fabsf(v.z)}; }
inline float maxcomp(V3 a) { return fps::fmax_(fps::fmax_(a.x, a.y), a.z); } // cutil_math.h:964

// common.h:17-24
const float PI = 3.14159265358f;
const float TWOPI = 6.28318530716f;
const float FOURPI = 12.56637061432f;
const float ONE_OVER_PI = 0.3183098861847f;
const float ONE_OVER_TWO_PI = 0.1591549430923f;
const float ONE_OVER_FOUR_PI = 0.0795774715461f;
// gamma(n), common.h:111-114: evaluated in double, returned as float.
inline float gamma_(int n) {
  const double me = (double)1.192092896e-07F * 0.5;
  return (float)((n * me) / (1 - n * me));
}

inline bool IsBlack(V3 c) { return c.x == 0 && c.y == 0 && c.z == 0; }  // common.h:72
inline bool IsNan(V3 c) { return c.x != c.x || c.y != c.y || c.z != c.z; }
inline bool isinf_(float v) { return v == INFINITY || v == -INFINITY; }
inline bool IsInf(V3 c) { return isinf_(c.x) || isinf_(c.y) || isinf_(c.z); }

inline void MakeCoordinate(V3 n, V3& u, V3& w) {  // common.h:116-129
  if (fabsf(n.x) > fabsf(n.y)) {
    float invLen = 1.0f / sqrtf(n.x * n.x + n.z * n.z);
    w = v3(n.z * invLen, 0.0f, -n.x * invLen);
  } else {
    float invLen = 1.0f / sqrtf(n.y * n.y + n.z * n.z);
    w = v3(0.0f, n.z * invLen, -n.y * invLen);
  }
  u = cross(w, n);
}
inline V3 ToWorld(V3 d, V3 u, V3 v, V3 w) { return d.x * u + d.y * v + d.z * w; }  // common.h:131

// ------------------------------------------------------------------- RNG
// common.h:200-209
inline uint32_t WangHash(uint32_t seed) {
  seed = (seed ^ 61u) ^ (seed >> 16);
  seed = seed + (seed << 3);
  seed = seed ^ (seed >> 4);
  seed = seed * 0x27d4eb2du;
  seed = seed ^ (seed >> 15);
  return seed;
}
// thrust::default_random_engine = minstd_rand = LCG(48271, 0, 2^31-1) and
// uniform_real_distribution<float>(0,1) (Thrust 1.9.x; same arithmetic in
// rocThrust's random/detail/linear_congruential_engine.inl:40-61 and
// uniform_real_distribution.inl:64-80).
struct Rng {
  uint32_t x;
  explicit Rng(uint32_t s) {
    x = s % 2147483647u;
    if (x == 0) x = 1;
  }
  uint32_t next() {
    x = (uint32_t)(((uint64_t)x * 48271u) % 2147483647u);
    return x;
  }
  float uniform() {
    float r = (float)(next() - 1u);
    r /= (1.0f + (float)2147483645u);
    return (r * (1.0f - 0.0f)) + 0.0f;
  }
};

// --------------------------------------------------------------- sampling
// Sampling.h:31-43 (the normal argument is unused there)
inline V3 CosineSampleHemiSphere(float u1, float u2, float& pdf) {
  float sintheta = sqrtf(u1);
  float costheta = sqrtf(1.f - u1);
  float phi = TWOPI * u2;
  float cosphi = fps::cos_(phi);
  float sinphi = fps::sin_(phi);
  pdf = fabsf(costheta) * ONE_OVER_PI;
  return v3(sintheta * cosphi, sintheta * sinphi, costheta);
}
// Sampling.h:4-15
inline V3 UniformSampleSphere(float u1, float u2, float& pdf) {
  float costheta = 1.f - 2.f * u1;
  float sintheta = sqrtf(1.f - costheta * costheta);
  float phi = TWOPI * u2;
  float cosphi = fps::cos_(phi);
  float sinphi = fps::sin_(phi);
  pdf = ONE_OVER_FOUR_PI;
  return v3(sintheta * cosphi, sintheta * sinphi, costheta);
}
// Sampling.h:59-67
inline V2 UniformSampleDisk(float u1, float u2) {
  float r = sqrtf(u1);
  float phi = TWOPI * u2;
  return V2{r * fps::cos_(phi), r * fps::sin_(phi)};
}
// Sampling.h:96-100
inline V2 UniformSampleTriangle(float u1, float u2) {
  float su1 = sqrtf(u1);
  return V2{1.f - su1, u2 * su1};
}

// ------------------------------------------------------------------ rays
struct Ray {  // ray.h:6-33
  V3 origin, dir;
  float tmin, tmax;
  int medium;  // index or -1 (the reference keeps a pointer)
  Ray() : origin{0, 0, 0}, dir{0, 0, 0}, tmin(0.001f), tmax(INFINITY), medium(-1) {}
  Ray(V3 o, V3 d, int m = -1, float mn = 0.001f, float mx = INFINITY)
      : origin(o), dir(d), tmin(mn), tmax(mx), medium(m) {}
  V3 at(float t) const { return origin + t * dir; }
};

struct Isect {  // intersection.h:4-18 (+ prim id)
  V3 pos{0, 0, 0}, n{0, 0, 0};
  V2 uv{0, 0};
  V3 dpdu{0, 0, 0};
  int matIdx = -1, lightIdx = -1, mediumInside = -1, mediumOutside = -1;
  int prim = -1;
};

enum RayKind { RK_EXT = 0, RK_SHADOW = 1, RK_MIS = 2, RK_TR = 3 };

struct Ctx {
  Scene scene;
  float g1, g2, g3;
};

struct Counters {
  oracle_counts c;
  Counters() { memset(&c, 0, sizeof(c)); }
};

// The two distances of the slab test (bbox.h:76-87): where the ray enters and leaves the box.
inline void SlabDistances(V3 lo, V3 hi, const Ray& r, float& tNear, float& tFar) {
  V3 inv = {1.f / r.dir.x, 1.f / r.dir.y, 1.f / r.dir.z};
  // (plane - origin) * inv as one fused multiply-add per plane where the ray allows it (fpspec.h)
  V3 noi = {-(r.origin.x * inv.x), -(r.origin.y * inv.y), -(r.origin.z * inv.z)};
  const bool fused = fps::slab_fused_(noi.x, noi.y, noi.z);
  float txNear = fps::slab_plane_(lo.x, r.origin.x, inv.x, noi.x, fused);
  float txFar = fps::slab_plane_(hi.x, r.origin.x, inv.x, noi.x, fused);
  float tyNear = fps::slab_plane_(lo.y, r.origin.y, inv.y, noi.y, fused);
  float tyFar = fps::slab_plane_(hi.y, r.origin.y, inv.y, noi.y, fused);
  float tzNear = fps::slab_plane_(lo.z, r.origin.z, inv.z, noi.z, fused);
  float tzFar = fps::slab_plane_(hi.z, r.origin.z, inv.z, noi.z, fused);
  using fps::fmax_;
  using fps::fmin_;
  tNear = fmax_(fmax_(fmin_(txNear, txFar), fmin_(tyNear, tyFar)), fmin_(tzNear, tzFar));
  tFar = fmin_(fmin_(fmax_(txNear, txFar), fmax_(tyNear, tyFar)), fmax_(tzNear, tzFar));
}

// bbox.h:75-98
inline bool BoxIntersect(const Ctx& cx, const Node& nd, const Ray& r) {
  float tNear, tFar;
  SlabDistances(v3(nd.lo), v3(nd.hi), r, tNear, tFar);
  tFar *= 1 + 2 * cx.g3;
  if (tFar <= 0.00001f) return false;
  if (tNear > tFar) return false;
  if (tNear > r.tmax * fps::kNearCullSlack) return false;  // (bbox.h:96 compares with tmax itself: see fpspec.h)
  return true;
}

// Triangle.h:87-144
inline bool TriIntersect(const Ctx& cx, const Prim& p, int id, Ray& ray, Isect* isect) {
  V3 p1 = v3(p.v[0].v), p2 = v3(p.v[1].v), p3 = v3(p.v[2].v);
  V3 e1 = p2 - p1;
  V3 e2 = p3 - p1;
  V3 s1 = cross(ray.dir, e2);
  float divisor = dot(s1, e1);
  if (fabsf(divisor) < 1e-8f) return false;
  float invDivisor = 1.0f / divisor;  // (float)(1.0/double(divisor)) rounds the same
  V3 s = ray.origin - p1;
  float b1 = dot(s, s1) * invDivisor;
  if (b1 < 0.0f || b1 > 1.0f) return false;
  V3 s2 = cross(s, e1);
  float b2 = dot(ray.dir, s2) * invDivisor;
  if (b2 < 0.0f || b1 + b2 > 1.0f) return false;

  float maxD = maxcomp(vabs(ray.dir));
  float maxE = fps::fmax_(fps::fmax_(maxcomp(vabs(p1)), maxcomp(vabs(p2))), maxcomp(vabs(p3)));
  float deltaE = cx.g1 * maxE;
  float deltaS1 = 2 * (cx.g2 * maxD * maxE + deltaE * maxD);
  float deltaS = cx.g1 * maxcomp(vabs(p1));
  float maxS = maxcomp(vabs(s));
  float maxS1 = maxcomp(vabs(s1));
  float deltaS2 = 2 * (cx.g2 * maxS * maxS1 + deltaS * maxS1 + deltaS1 * maxS);
  float maxS2 = maxcomp(vabs(s2));
  float deltaT = 2 * (cx.g2 * maxE * maxS2 + deltaS2 * maxE + deltaE * maxS2) * fabsf(invDivisor);

  float tt = dot(e2, s2) * invDivisor;
  if (tt < ray.tmin + deltaT || tt > ray.tmax) return false;

  ray.tmax = tt;
  if (isect) {
    V3 ns = normalize(v3(p.v[0].n) * (1.f - b1 - b2) + v3(p.v[1].n) * b1 + v3(p.v[2].n) * b2);
    V3 ss = normalize(v3(p.v[0].t) * (1.f - b1 - b2) + v3(p.v[1].t) * b1 + v3(p.v[2].t) * b2);
    V3 ts = cross(ss, ns);
    if (length(ts) > 0.f) {
      ts = normalize(ts);
      ss = cross(ts, ns);
    }
    isect->pos = ray.at(tt);
    isect->n = ns;
    V2 u1 = {p.v[0].uv.x, p.v[0].uv.y}, u2 = {p.v[1].uv.x, p.v[1].uv.y}, u3 = {p.v[2].uv.x, p.v[2].uv.y};
    float b0 = 1.f - b1 - b2;
    isect->uv = V2{u1.x * b0 + u2.x * b1 + u3.x * b2, u1.y * b0 + u2.y * b1 + u3.y * b2};
    isect->dpdu = ss;
    isect->matIdx = p.mat;
    isect->lightIdx = p.light;
    isect->mediumInside = p.med_in;
    isect->mediumOutside = p.med_out;
    isect->prim = id;
  }
  return true;
}

// sphere.h:24-91
inline bool SphereIntersect(const Prim& p, int id, Ray& ray, Isect* isect, uint64_t* box_rejects = nullptr) {
  V3 origin = v3(p.p0);
  float radius = p.w0;
  V3 op = ray.origin - origin;
  float B = dot(op, ray.dir);
  float C = dot(op, op) - radius * radius;
  float delta = B * B - C;
  if (delta < 0.f) return false;
  float sqrDelta = sqrtf(delta);
  float t1 = -B - sqrDelta;
  float t2 = -B + sqrDelta;
  if (t1 < 0.f && t2 < 0.f) return false;
  if (t1 < 0.f || t2 < 0.f) {
    float tt1 = t1, tt2 = t2;
    t1 = tt1 < 0.f ? tt2 : tt1;
    t2 = tt1 < 0.f ? tt1 : tt2;
  } else if (t1 > t2) {
    float temp = t2; t2 = t1; t1 = temp;
  }
  if (t1 > ray.tmax) return false;
  float th;
  if (t1 > ray.tmin) th = t1;
  else if (t2 > 0.f) th = t2;
  else return false;
  if (th > ray.tmax) return false;  // fixed convention, see file header
  {
    // Fixed convention (file header): the roots above are distances only for a unit direction.
    // A candidate that lies before the point where the ray enters the sphere's own bounding box
    // (sphere.h:15-18, the box the tree encloses) is no hit.
    V3 rr = v3(radius, radius, radius);
    float tNear, tFar;
    SlabDistances(origin - rr, origin + rr, ray, tNear, tFar);
    if (tNear > th * fps::kNearCullSlack) {
      if (box_rejects) ++*box_rejects;
      return false;
    }
  }
  ray.tmax = th;
  if (isect) {
    isect->pos = ray.at(ray.tmax);
    isect->n = normalize(isect->pos - origin);
    V3 normal = isect->n;
    float costheta = dot(normal, v3(0.f, 1.f, 0.f));
    float v = fps::acos_(costheta) * ONE_OVER_PI;
    float cosphi = dot(v3(1.f, 0.f, 0.f), v3(normal.x, 0.f, normal.z));
    float phi = fps::acos_(cosphi);
    phi = normal.z > 0.f ? TWOPI - phi : phi;
    float u = phi * ONE_OVER_TWO_PI;
    isect->dpdu = normalize(v3(-TWOPI * isect->pos.y, TWOPI * isect->pos.x, 0));
    isect->uv = V2{u, v};
    isect->matIdx = p.mat;
    isect->lightIdx = -1;
    isect->mediumInside = p.med_in;
    isect->mediumOutside = p.med_out;
    isect->prim = id;
  }
  return true;
}

// line.h:33-87
inline bool LineIntersect(const Prim& p, int id, Ray& ray, Isect* isect) {
  V3 p0 = v3(p.p0), p1 = v3(p.p1);
  V3 u = ray.dir;
  V3 v = p1 - p0;
  V3 w = ray.origin - p0;
  float a = dot(u, u), b = dot(u, v), c = dot(v, v), d = dot(u, w), e = dot(v, w);
  float det = a * c - b * b;
  if (det == 0) return false;
  float t = (b * e - c * d) / det;
  float s = (a * e - b * d) / det;
  if (t < ray.tmin || t > ray.tmax) return false;
  s = fps::clamp_(s, 0.f, 1.f);
  V3 pr = ray.origin + ray.dir * t;
  V3 pl = p0 + (p1 - p0) * s;
  V3 prl = pr - pl;
  float d2 = dot(prl, prl);
  float r = p.w0 * (1 - s) + p.w1 * s;
  if (d2 > r * r) return false;
  ray.tmax = t;
  if (isect) {
    isect->pos = ray.at(t);
    isect->n = -ray.dir;
    isect->uv = V2{s, sqrtf(d2) / r};
    V3 dpdu, dpdv;
    MakeCoordinate(isect->n, dpdu, dpdv);
    isect->dpdu = dpdu;
    isect->matIdx = p.mat;
    isect->lightIdx = -1;
    isect->mediumInside = -1;
    isect->mediumOutside = -1;
    isect->prim = id;
  }
  return true;
}

inline bool PrimIntersect(const Ctx& cx, int i, Ray& ray, Isect* isect, Counters& k, int kind) {
  const Prim& p = cx.scene.prims[i];
  if (p.type == caster::GT_TRIANGLE) { k.c.tris[kind]++; return TriIntersect(cx, p, i, ray, isect); }
  if (p.type == caster::GT_LINES) { k.c.lines[kind]++; return LineIntersect(p, i, ray, isect); }
  k.c.spheres[kind]++;
  return SphereIntersect(p, i, ray, isect, &k.c.sphere_box_rejects);
}

// Test aid (oracle_debug_bruteforce): every ray is also tested against all primitives, and a walk
// of the tree that finds another answer is reported on stderr.
static bool g_debug_brute = false;
static void DebugBrute(const Ctx& cx, const Ray& ray0, bool any, bool got, int got_prim, float got_t, int kind) {
  Counters scratch;
  Ray r = ray0;
  Isect is;
  bool hit = false;
  fprintf(stderr, "RAY %d %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", kind, ray0.origin.x, ray0.origin.y, ray0.origin.z, ray0.dir.x,
          ray0.dir.y, ray0.dir.z, ray0.tmax);
  float len2 = dot(ray0.dir, ray0.dir);
  if (!(len2 > 0.999f && len2 < 1.001f))
    fprintf(stderr, "oracle: ray kind %d has a direction of length^2 %.9g\n", kind, len2);
  for (int i = 0; i < (int)cx.scene.prims.size(); ++i)
    if (PrimIntersect(cx, i, r, any ? nullptr : &is, scratch, kind)) { hit = true; if (any) break; }
  if (hit != got || (!any && hit && is.prim != got_prim))
    fprintf(stderr, "oracle: %s ray kind %d o=(%.9g %.9g %.9g) d=(%.9g %.9g %.9g) tmax=%.9g: tree %d (prim %d t %.9g), all primitives %d (prim %d t %.9g)\n",
            any ? "any-hit" : "closest-hit", kind, ray0.origin.x, ray0.origin.y, ray0.origin.z, ray0.dir.x, ray0.dir.y, ray0.dir.z,
            ray0.tmax, (int)got, got_prim, got_t, (int)hit, any ? -1 : is.prim, any ? 0.f : r.tmax);
  if (hit != got || (!any && hit && is.prim != got_prim)) {
    // the chain of boxes above the primitive the walk did not reach, each with its slab test
    int want = any ? -1 : is.prim;
    if (want < 0) return;
    for (size_t n = 0; n < cx.scene.nodes.size(); ++n) {
      const Node& nd = cx.scene.nodes[n];
      if (nd.leaf && nd.start <= want && want <= nd.end) {
        const Prim& pr = cx.scene.prims[want];
        fprintf(stderr, "  primitive %d is in leaf %zu [%d..%d] box (%.9g %.9g %.9g)-(%.9g %.9g %.9g); vertices (%.9g %.9g %.9g) (%.9g %.9g %.9g) (%.9g %.9g %.9g)\n",
                want, n, nd.start, nd.end, nd.lo.x, nd.lo.y, nd.lo.z, nd.hi.x, nd.hi.y, nd.hi.z, pr.v[0].v.x, pr.v[0].v.y, pr.v[0].v.z,
                pr.v[1].v.x, pr.v[1].v.y, pr.v[1].v.z, pr.v[2].v.x, pr.v[2].v.y, pr.v[2].v.z);
      }
    }
    int idx = 0;
    for (;;) {
      const Node& nd = cx.scene.nodes[idx];
      Ray rr = ray0;
      V3 inv = {1.f / rr.dir.x, 1.f / rr.dir.y, 1.f / rr.dir.z};
      float tx0 = (nd.lo.x - rr.origin.x) * inv.x, tx1 = (nd.hi.x - rr.origin.x) * inv.x;
      float ty0 = (nd.lo.y - rr.origin.y) * inv.y, ty1 = (nd.hi.y - rr.origin.y) * inv.y;
      float tz0 = (nd.lo.z - rr.origin.z) * inv.z, tz1 = (nd.hi.z - rr.origin.z) * inv.z;
      fprintf(stderr, "  node %d leaf %d box (%.9g %.9g %.9g)-(%.9g %.9g %.9g): x [%.9g %.9g] y [%.9g %.9g] z [%.9g %.9g] -> %d\n", idx, (int)nd.leaf,
              nd.lo.x, nd.lo.y, nd.lo.z, nd.hi.x, nd.hi.y, nd.hi.z, tx0, tx1, ty0, ty1, tz0, tz1, (int)BoxIntersect(cx, nd, rr));
      if (nd.leaf) break;
      // pre-order layout: the first child follows its parent, the second starts at nd.second
      // and holds the primitives from its first leaf on
      int probe = nd.second;
      while (!cx.scene.nodes[probe].leaf) probe = probe + 1;
      idx = want >= cx.scene.nodes[probe].start ? nd.second : idx + 1;
    }
  }
}

// pathtracer.cu:223-264 (closest hit; unordered depth-first walk)
bool Intersect(const Ctx& cx, Ray& ray, Isect* isect, Counters& k, int kind) {
  const Ray ray0 = ray;
  k.c.rays[kind]++;
  if (cx.scene.nodes.empty()) return false;
  // fixed convention (file header): no hit is looked for beyond the scene's far distance
  ray.tmax = ray.tmax < cx.scene.far_t ? ray.tmax : cx.scene.far_t;
  std::vector<int>& stack = *([]() { static thread_local std::vector<int> s(256); return &s; })();
  int top = 0;
  bool ret = false;
  int node_idx = 0;
  for (;;) {
    const Node& node = cx.scene.nodes[node_idx];
    k.c.nodes[kind]++;
    if (BoxIntersect(cx, node, ray)) {
      if (!node.leaf) {
        if (top + 2 > (int)stack.size()) stack.resize(stack.size() * 2);
        stack[top++] = node.second;
        stack[top++] = node_idx + 1;
      } else {
        for (int i = node.start; i <= node.end && node.start >= 0; ++i)
          if (PrimIntersect(cx, i, ray, isect, k, kind)) ret = true;
      }
    }
    if (top == 0) break;
    node_idx = stack[--top];
  }
  if (ret) k.c.hits[kind]++;
  else ray.tmax = ray0.tmax;   // (a ray that found nothing keeps the bound it came with)
  if (g_debug_brute) DebugBrute(cx, ray0, false, ret, ret && isect ? isect->prim : -1, ray.tmax, kind);
  return ret;
}

// pathtracer.cu:266-305 (any hit)
static bool IntersectPrimitiveWalk(const Ctx& cx, Ray& ray, Counters& k, int kind);
bool IntersectPrimitive(const Ctx& cx, Ray& ray, Counters& k, int kind) {
  if (!g_debug_brute) return IntersectPrimitiveWalk(cx, ray, k, kind);
  const Ray ray0 = ray;
  bool got = IntersectPrimitiveWalk(cx, ray, k, kind);
  DebugBrute(cx, ray0, true, got, -1, 0.f, kind);
  return got;
}
static bool IntersectPrimitiveWalk(const Ctx& cx, Ray& ray, Counters& k, int kind) {
  k.c.rays[kind]++;
  if (cx.scene.nodes.empty()) return false;
  const float tmax0 = ray.tmax;
  ray.tmax = ray.tmax < cx.scene.far_t ? ray.tmax : cx.scene.far_t;   // fixed convention (file header)
  std::vector<int>& stack = *([]() { static thread_local std::vector<int> s(256); return &s; })();
  int top = 0;
  int node_idx = 0;
  for (;;) {
    const Node& node = cx.scene.nodes[node_idx];
    k.c.nodes[kind]++;
    if (BoxIntersect(cx, node, ray)) {
      if (!node.leaf) {
        if (top + 2 > (int)stack.size()) stack.resize(stack.size() * 2);
        stack[top++] = node.second;
        stack[top++] = node_idx + 1;
      } else {
        for (int i = node.start; i <= node.end && node.start >= 0; ++i)
          if (PrimIntersect(cx, i, ray, nullptr, k, kind)) { k.c.hits[kind]++; return true; }
      }
    }
    if (top == 0) break;
    node_idx = stack[--top];
  }
  ray.tmax = tmax0;
  return false;
}

// ------------------------------------------------------------------ media
const int kMaxCrossings = 1024;

inline V3 Exp(V3 c) { return v3(fps::exp_(c.x), fps::exp_(c.y), fps::exp_(c.z)); }  // common.h:84

// medium.h:13-16
inline V3 HomogeneousTr(const caster::Medium& m, const Ray& ray) {
  V3 c = v3(m.sigmaT) * (-ray.tmax);
  return Exp(c);
}
// medium.h:18-38
inline V3 HomogeneousSample(const caster::Medium& m, const Ray& ray, Rng& rng, float& t, bool& sampled) {
  int channel = fps::f2i_(rng.uniform() * 3);
  channel = channel == 3 ? 2 : channel;
  float sT[3] = {m.sigmaT.x, m.sigmaT.y, m.sigmaT.z};
  float dist = -fps::log_(1 - rng.uniform()) / sT[channel];
  dist = dist < ray.tmax ? dist : ray.tmax;
  sampled = dist < ray.tmax;
  if (sampled) t = dist;
  V3 Tr = Exp(v3(m.sigmaT) * -dist);
  V3 density = sampled ? (v3(m.sigmaT) * Tr) : Tr;
  float pdf = 0;
  pdf += density.x; pdf += density.y; pdf += density.z;
  pdf /= 3.f;
  if (pdf == 0) pdf = 1;
  return sampled ? (Tr * v3(m.sigmaS) / pdf) : (Tr / pdf);
}
// medium.h:116-129: grid lookup, zero outside; float -> int truncates
inline float HetCell(const caster::Medium& m, V3 p) {
  int x = fps::f2i_(p.x), y = fps::f2i_(p.y), z = fps::f2i_(p.z);
  if (x < 0 || x > m.nx - 1 || y < 0 || y > m.ny - 1 || z < 0 || z > m.nz - 1) return 0.f;
  return m.density[(size_t)z * m.ny * m.nx + (size_t)y * m.nx + x];
}
inline float Lerp(float a, float b, float t) { return a + t * (b - a); }  // cutil_math.h lerp
// medium.h:103-114: trilinear density at p in [0,1]^3 of the grid's box
inline float HetDensity(const caster::Medium& m, V3 p) {
  V3 ps = v3(p.x * m.nx - .5f, p.y * m.ny - .5f, p.z * m.nz - .5f);
  V3 psi = v3(floorf(ps.x), floorf(ps.y), floorf(ps.z));
  V3 delta = ps - psi;
  float d00 = Lerp(HetCell(m, psi), HetCell(m, psi + v3(1, 0, 0)), delta.x);
  float d10 = Lerp(HetCell(m, psi + v3(0, 1, 0)), HetCell(m, psi + v3(1, 1, 0)), delta.x);
  float d01 = Lerp(HetCell(m, psi + v3(0, 0, 1)), HetCell(m, psi + v3(1, 0, 1)), delta.x);
  float d11 = Lerp(HetCell(m, psi + v3(0, 1, 1)), HetCell(m, psi + v3(1, 1, 1)), delta.x);
  float d0 = Lerp(d00, d10, delta.y);
  float d1 = Lerp(d01, d11, delta.y);
  return Lerp(d0, d1, delta.z);
}
// medium.h:53-79: ratio tracking
inline V3 HeterogeneousTr(const caster::Medium& m, const Ray& ray, Rng& rng) {
  float sigma = dot(v3(m.sigmaT), v3(0.212671f, 0.715160f, 0.072169f));
  V3 d = v3(m.p1) - v3(m.p0);
  float tr = 1.f;
  float dist = 0.f;
  int iter = m.iterMax;
  while (true) {
    dist += -fps::log_(rng.uniform()) * m.invMaxDensity / sigma;
    if (dist >= ray.tmax) break;
    V3 p = ray.at(dist);
    p = (p - v3(m.p0)) / d;
    tr *= 1.f - HetDensity(m, p) * m.invMaxDensity;
    if (tr < 0.1f) {
      float q = 1.f - tr;
      if (rng.uniform() < q) return v3(0.f, 0.f, 0.f);
      tr /= (1.f - q);
    }
    if (--iter == 0) break;
  }
  return v3(tr, tr, tr);
}
// medium.h:81-109: delta tracking
inline V3 HeterogeneousSample(const caster::Medium& m, const Ray& ray, Rng& rng, float& t, bool& sampled) {
  float sigma = dot(v3(m.sigmaT), v3(0.212671f, 0.715160f, 0.072169f));
  V3 d = v3(m.p1) - v3(m.p0);
  float dist = 0.f;
  int iter = m.iterMax;
  while (true) {
    dist += -fps::log_(rng.uniform()) * m.invMaxDensity / sigma;
    if (dist >= ray.tmax) break;
    V3 p = ray.at(dist);
    p = (p - v3(m.p0)) / d;
    if (HetDensity(m, p) * m.invMaxDensity > rng.uniform()) {
      t = dist;
      sampled = true;
      return v3(m.sigmaS) / v3(m.sigmaT);
    }
    if (--iter == 0) break;
  }
  t = dist;
  sampled = false;
  return v3(1.f, 1.f, 1.f);
}
// the type switch of pathtracer.cu:317-320, 1056-1059
inline V3 MediumTr(const caster::Medium& m, const Ray& ray, Rng& rng) {
  return m.type == caster::MED_HOMOGENEOUS ? HomogeneousTr(m, ray) : HeterogeneousTr(m, ray, rng);
}
inline V3 MediumSample(const caster::Medium& m, const Ray& ray, Rng& rng, float& t, bool& sampled) {
  return m.type == caster::MED_HOMOGENEOUS ? HomogeneousSample(m, ray, rng, t, sampled)
                                           : HeterogeneousSample(m, ray, rng, t, sampled);
}
// medium.h:146-172
inline void SamplePhase(const caster::Medium& m, V2 u, V3& dir, float& phase, float& pdf) {
  float g = m.g;
  if (g == 0) {
    phase = ONE_OVER_FOUR_PI;
    dir = UniformSampleSphere(u.x, u.y, pdf);
    return;
  }
  float costheta;
  if (fabsf(g) < 1e-3f) costheta = 1.f - 2.f * u.x;
  else {
    float sqrtTerm = (1.f - g * g) / (1.f - g + 2.f * g * u.x);
    costheta = (1.f + g * g - sqrtTerm * sqrtTerm) / (2.f * g);
  }
  float sintheta = sqrtf(1.f - costheta * costheta);
  float phi = TWOPI * u.y;
  float sinphi = fps::sin_(phi), cosphi = fps::cos_(phi);
  dir = v3(sintheta * cosphi, sintheta * sinphi, costheta);
  float cubicTerm = (1.f + g * g - 2.f * g * costheta);
  phase = ONE_OVER_FOUR_PI * (1.f - g * g) / (cubicTerm * sqrtf(cubicTerm));
  pdf = phase;
}
// medium.h:174-186
inline void Phase(const caster::Medium& m, V3 in, V3 out, float& phase, float& pdf) {
  float g = m.g;
  if (g == 0) { phase = ONE_OVER_FOUR_PI; pdf = phase; return; }
  float costheta = dot(in, out);
  float cubicTerm = (1.f + g * g - 2.f * g * costheta);
  phase = ONE_OVER_FOUR_PI * (1.f - g * g) / sqrtf(cubicTerm * cubicTerm * cubicTerm);
  pdf = phase;
}

inline int MediumBySide(float side, const Isect& is) {  // pathtracer.cu:324-325
  return side > 0 ? is.mediumOutside : is.mediumInside;
}

// pathtracer.cu:307-331
V3 Tr(const Ctx& cx, Ray& ray, Rng& rng, Counters& k) {
  V3 tr = v3(1, 1, 1);
  float tmax = ray.tmax;
  for (int guard = 0; guard < kMaxCrossings; ++guard) {
    Isect isect;
    bool invisible = Intersect(cx, ray, &isect, k, RK_TR);
    if (invisible && isect.matIdx != -1) return v3(0, 0, 0);
    if (ray.medium != -1) tr *= MediumTr(cx.scene.media[ray.medium], ray, rng);
    if (!invisible) break;
    int m = MediumBySide(dot(ray.dir, isect.n), isect);
    tmax -= ray.tmax;
    ray = Ray(ray.at(ray.tmax), ray.dir, m, cx.scene.epsilon, tmax);
  }
  return tr;
}

// --------------------------------------------------------------- textures
// pathtracer.cu:333-348
inline V4 getTexel(const caster::Texture& tx, int x, int y) {
  int w = tx.w, h = tx.h;
  float inv = 1.f / 255.f;
  float rx = (float)(x - (x / w) * w);
  float ry = (float)(y - (y / h) * h);
  x = fps::f2i_((rx < 0) ? rx + w : rx);
  y = fps::f2i_((ry < 0) ? ry + h : ry);
  if (x < 0) x = 0;
  if (x > w - 1) x = w - 1;
  if (y < 0) y = 0;
  if (y > h - 1) y = h - 1;
  const uint8_t* c = &tx.rgba[((size_t)y * w + x) * 4];
  return V4{c[0] * inv, c[1] * inv, c[2] * inv, c[3] * inv};
}
// pathtracer.cu:350-368
inline V3 GetTexel(const Ctx& cx, const Material& m, V2 uv) {
  if (m.tex == -1) return v3(m.diffuse);
  const caster::Texture& tx = cx.scene.textures[m.tex];
  float xx = tx.w * uv.x;
  float yy = tx.h * uv.y;
  int x = fps::f2i_(floorf(xx));
  int y = fps::f2i_(floorf(yy));
  float dx = fabsf(xx - x);
  float dy = fabsf(yy - y);
  V4 c00 = getTexel(tx, x, y), c10 = getTexel(tx, x + 1, y);
  V4 c01 = getTexel(tx, x, y + 1), c11 = getTexel(tx, x + 1, y + 1);
  V4 r = (1 - dy) * ((1 - dx) * c00 + dx * c10) + dy * ((1 - dx) * c01 + dx * c11);
  return v3(r.x, r.y, r.z);
}

// ------------------------------------------------------------------ BSDFs
// pathtracer.cu:45-48
inline V3 SphericalDirection(float s, float c, float phi) {
  return v3(s * fps::cos_(phi), s * fps::sin_(phi), c);
}
// pathtracer.cu:50-56
inline float DielectricFresnel(float cosi, float cost, float etai, float etat) {
  float Rparl = (etat * cosi - etai * cost) / (etat * cosi + etai * cost);
  float Rperp = (etai * cosi - etat * cost) / (etai * cosi + etat * cost);
  return (Rparl * Rparl + Rperp * Rperp) * 0.5f;
}
// pathtracer.cu:58-66
inline V3 ConductFresnel(float cosi, V3 eta, V3 k) {
  V3 tmp = (eta * eta + k * k) * cosi * cosi;
  V3 Rparl2 = (tmp - eta * cosi * 2.f + 1.f) / (tmp + eta * cosi * 2.f + 1.f);
  V3 tmp_f = (eta * eta + k * k);
  V3 Rperp2 = (tmp_f - eta * cosi * 2.f + cosi * cosi) / (tmp_f + eta * cosi * 2.f + cosi * cosi);
  return (Rparl2 + Rperp2) * 0.5f;
}
// pathtracer.cu:68-82 (keeps the (1-cos phi) term as written)
inline float GGX_D(V3 wh, V3 normal, V3 dpdu, float alphaU, float alphaV) {
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
// pathtracer.cu:106-121
inline float SmithG(V3 w, V3 normal, V3 wh, V3 dpdu, float alphaU, float alphaV) {
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
// pathtracer.cu:123-125
inline float GGX_G(V3 wo, V3 wi, V3 normal, V3 wh, V3 dpdu, float aU, float aV) {
  return SmithG(wo, normal, wh, dpdu, aU, aV) * SmithG(wi, normal, wh, dpdu, aU, aV);
}
// pathtracer.cu:127-147 (anisotropic branch keeps alphaU2 twice)
inline V3 SampleGGX(float alphaU, float alphaV, float u1, float u2) {
  if (alphaU == alphaV) {
    float costheta = sqrtf((1.f - u1) / (u1 * (alphaU * alphaV - 1.f) + 1.f));
    float sintheta = sqrtf(1.f - costheta * costheta);
    float phi = TWOPI * u2;
    return SphericalDirection(sintheta, costheta, phi);
  }
  float phi = fps::atan_(alphaV / alphaU * fps::tan_(TWOPI * u2 + .5f * PI));
  if (u2 > .5f) phi += PI;
  float sinphi = fps::sin_(phi), cosphi = fps::cos_(phi);
  const float alphaU2 = alphaU * alphaU;
  const float alpha2 = 1.0f / (cosphi * cosphi / alphaU2 + sinphi * sinphi / alphaU2);
  float tanTheta2 = alpha2 * u1 / (1.f - u1);
  float costheta = 1 / sqrtf(1 + tanTheta2);
  float sintheta = sqrtf(fps::fmax_(0.f, 1.f - costheta * costheta));
  return SphericalDirection(sintheta, costheta, phi);
}
inline V3 Reflect(V3 in, V3 nor) { return 2.f * dot(in, nor) * nor - in; }  // pathtracer.cu:149
// pathtracer.cu:153-167
inline V3 Refract(V3 in, V3 nor, float etai, float etat) {
  float cosi = dot(in, nor);
  bool enter = cosi > 0;
  if (!enter) { float t = etai; etai = etat; etat = t; }
  float eta = etai / etat;
  float sini2 = 1.f - cosi * cosi;
  float sint2 = sini2 * eta * eta;
  float cost = sqrtf(1.f - sint2);
  return normalize((nor * cosi - in) * eta + (enter ? -cost : cost) * nor);
}
inline float pow5(float v) { return (v * v) * (v * v) * v; }
// pathtracer.cu:169-173
inline V3 SchlickFresnel(V3 Rs, float cosTheta) { return Rs + pow5(1.f - cosTheta) * (1 - Rs); }
// pathtracer.cu:175-178
inline float PowerHeuristic(int nf, float fPdf, int ng, float gPdf) {
  float f = nf * fPdf, g = ng * gPdf;
  return (f * f) / (f * f + g * g);
}
inline float Luminance(V3 c) { return dot(c, v3(0.212671f, 0.715160f, 0.072169f)); }  // :215
inline bool SameHemiSphere(V3 in, V3 out, V3 nor) { return dot(in, nor) * dot(out, nor) > 0; }  // :219
inline bool IsDelta(int t) { return t == caster::MT_MIRROR || t == caster::MT_DIELECTRIC; }  // material.h:33

// pathtracer.cu:500-698; `radiance` false is TransportMode::Importance (light paths), which drops
// the two radiance-scaling factors of refraction (:549-550, :683-684)
void SampleBSDF(const Ctx& cx, const Material& material, V3 in, V3 nor, V2 uv, V3 dpdu, V3 u, V3& out,
                V3& fr, float& pdf, bool radiance = true) {
  switch (material.type) {
    case caster::MT_LAMBERTIAN: {
      V3 n = nor;
      if (dot(nor, in) < 0) n = -n;
      out = CosineSampleHemiSphere(u.x, u.y, pdf);
      V3 uu = dpdu, ww = cross(uu, n);
      out = ToWorld(out, uu, ww, n);
      fr = GetTexel(cx, material, uv) * ONE_OVER_PI;
      break;
    }
    case caster::MT_MIRROR:
      out = Reflect(in, nor);
      fr = v3(material.specular) / fabsf(dot(out, nor));
      pdf = 1.f;
      break;
    case caster::MT_DIELECTRIC: {
      V3 wi = -in;
      V3 normal = nor;
      float ei = material.outsideIOR, et = material.insideIOR;
      float cosi = dot(wi, normal);
      if (cosi > 0) { float t = et; et = ei; ei = t; }
      float eta = ei / et;
      float sint2 = eta * eta * (1.f - cosi * cosi);
      float cost = sqrtf(1.f - sint2 < 0.f ? 0.f : 1.f - sint2);
      V3 rdir = Reflect(-wi, normal);
      V3 tdir = Refract(in, nor, material.outsideIOR, material.insideIOR);
      if (sint2 > 1.f) {
        out = rdir;
        fr = v3(material.specular) / fabsf(dot(out, normal));
        pdf = 1.f;
        return;
      }
      float fresnel = DielectricFresnel(fabsf(cost), fabsf(cosi), et, ei);
      if (u.x > fresnel) {
        out = tdir;
        fr = v3(material.specular) / fabsf(dot(out, normal)) * (1.f - fresnel);
        if (radiance) fr *= eta * eta;
        pdf = 1.f - fresnel;
      } else {
        out = rdir;
        fr = v3(material.specular) / fabsf(dot(out, normal)) * fresnel;
        pdf = fresnel;
      }
      break;
    }
    case caster::MT_ROUGHCONDUCTOR: {
      V3 n = nor;
      if (dot(nor, in) < 0) n = -n;
      V3 wh = SampleGGX(material.alphaU, material.alphaV, u.x, u.y);
      V3 uu = dpdu, ww = cross(uu, n);
      wh = ToWorld(wh, uu, ww, n);
      out = Reflect(in, wh);
      if (!SameHemiSphere(in, out, nor)) { fr = v3(0, 0, 0); pdf = 0.f; return; }
      float cosi = dot(out, wh);
      V3 F = ConductFresnel(fabsf(cosi), v3(material.eta), v3(material.k));
      float D = GGX_D(wh, n, dpdu, material.alphaU, material.alphaV);
      float G = GGX_G(in, out, n, wh, dpdu, material.alphaU, material.alphaV);
      fr = v3(material.specular) * F * D * G / (4.f * fabsf(dot(in, n)) * fabsf(dot(out, n)));
      pdf = D * fabsf(dot(wh, n)) / (4.f * fabsf(dot(in, wh)));
      break;
    }
    case caster::MT_SUBSTRATE: {
      V3 n = nor;
      if (dot(nor, in) < 0) n = -n;
      if (u.x < 0.5f) {
        u.x = fps::fmin_(2.f * u.x, 0x1.fffffep-1f);
        out = CosineSampleHemiSphere(u.x, u.y, pdf);
        V3 uu = dpdu, ww = cross(uu, n);
        out = ToWorld(out, uu, ww, n);
      } else {
        u.x = fps::fmin_(2 * (u.x - 0.5f), 0x1.fffffep-1f);
        V3 wh = SampleGGX(material.alphaU, material.alphaV, u.x, u.y);
        V3 uu = dpdu, ww = cross(uu, n);
        wh = ToWorld(wh, uu, ww, n);
        out = Reflect(in, wh);
      }
      if (!SameHemiSphere(in, out, n)) { fr = v3(0, 0, 0); pdf = 0.f; return; }
      float cosThetaWo = fabsf(dot(in, n));
      float cosThetaWi = fabsf(dot(out, n));
      V3 Rd = GetTexel(cx, material, uv);
      V3 Rs = v3(material.specular);
      V3 diffuse = (28.f / (23.f * PI)) * Rd * (v3(1.f, 1.f, 1.f) - Rs) *
                   (1 - pow5(1 - .5f * cosThetaWi)) * (1 - pow5(1 - .5f * cosThetaWo));
      V3 wh = normalize(in + out);
      float D = GGX_D(wh, n, dpdu, material.alphaU, material.alphaV);
      float mx = cosThetaWi > cosThetaWo ? cosThetaWi : cosThetaWo;  // Max<T>, common.h:98-101
      V3 specular = D / (4.f * fabsf(dot(out, wh)) * mx) * SchlickFresnel(Rs, dot(out, wh));
      fr = diffuse + specular;
      pdf = 0.5f * (fabsf(dot(out, n)) * ONE_OVER_PI + D * fabsf(dot(wh, n)) / (4.f * dot(in, wh)));
      break;
    }
    case caster::MT_ROUGHDIELECTRIC: {
      V3 wo = -in;
      V3 n = nor;
      V3 wh = SampleGGX(material.alphaU, material.alphaV, u.x, u.y);
      V3 uu = dpdu, ww = cross(uu, n);
      wh = ToWorld(wh, uu, ww, n);
      float ei = material.outsideIOR, et = material.insideIOR;
      float cosi = dot(wo, n);
      if (cosi > 0) { float t = et; et = ei; ei = t; }
      float eta = ei / et;
      float D = GGX_D(wh, n, dpdu, material.alphaU, material.alphaV);
      float cosThetaI = dot(wo, wh);
      float sin2ThetaI = fps::fmax_(0.f, 1.f - cosThetaI * cosThetaI);
      float sin2ThetaT = eta * eta * sin2ThetaI;
      float cosThetaT = sqrtf(1.f - sin2ThetaT < 0.f ? 0.f : 1.f - sin2ThetaT);
      V3 rdir = Reflect(-wo, wh);
      V3 tdir = normalize((wo - wh * cosThetaI) * eta + (cosi < 0 ? -cosThetaT : cosThetaT) * wh);
      if (sin2ThetaT > 1.f) {
        out = rdir;
        float G = GGX_G(in, out, n, wh, dpdu, material.alphaU, material.alphaV);
        fr = v3(material.specular) * D * G / (4.f * fabsf(dot(in, n)) * fabsf(dot(out, n)));
        pdf = D * fabsf(dot(wh, n)) / (4.f * fabsf(dot(wh, in)));
        return;
      }
      float fresnel = DielectricFresnel(fabsf(cosThetaI), fabsf(cosThetaT), ei, et);
      if (u.z > fresnel) {
        out = tdir;
        float G = GGX_G(in, out, n, wh, dpdu, material.alphaU, material.alphaV);
        float c = et * dot(out, wh) + ei * dot(in, wh);
        fr = v3(material.specular) * ei * ei * D * G * (1.f - fresnel) * fabsf(dot(in, wh)) *
             fabsf(dot(out, wh)) / (fabsf(dot(out, n)) * fabsf(dot(in, n)) * c * c);
        if (radiance) fr *= (1.f / (eta * eta));
        pdf = (1.f - fresnel) * D * fabsf(dot(wh, n)) * et * et * fabsf(dot(out, wh)) / (c * c);
      } else {
        out = rdir;
        float G = GGX_G(in, out, n, wh, dpdu, material.alphaU, material.alphaV);
        fr = v3(material.specular) * fresnel * D * G / (4.f * fabsf(dot(in, n)) * fabsf(dot(out, n)));
        pdf = D * fabsf(dot(wh, n)) / (4.f * fabsf(dot(wh, in))) * fresnel;
      }
      break;
    }
  }
}

// pathtracer.cu:700-818
void Fr(const Ctx& cx, const Material& material, V3 in, V3 out, V3 nor, V2 uv, V3 dpdu, V3& fr, float& pdf) {
  switch (material.type) {
    case caster::MT_LAMBERTIAN:
      if (!SameHemiSphere(in, out, nor)) { fr = v3(0, 0, 0); pdf = 0.f; return; }
      fr = GetTexel(cx, material, uv) * ONE_OVER_PI;
      pdf = fabsf(dot(out, nor)) * ONE_OVER_PI;
      break;
    case caster::MT_MIRROR:
    case caster::MT_DIELECTRIC:
      fr = v3(0, 0, 0);
      pdf = 0.f;
      break;
    case caster::MT_ROUGHCONDUCTOR: {
      if (!SameHemiSphere(in, out, nor)) { fr = v3(0, 0, 0); pdf = 0; return; }
      V3 n = nor;
      if (dot(nor, in) < 0) n = -n;
      V3 wh = normalize(in + out);
      float cosi = dot(out, wh);
      float D = GGX_D(wh, n, dpdu, material.alphaU, material.alphaV);
      float G = GGX_G(in, out, n, wh, dpdu, material.alphaU, material.alphaV);
      V3 F = ConductFresnel(fabsf(cosi), v3(material.eta), v3(material.k));
      fr = v3(material.specular) * F * D * G / (4.f * fabsf(dot(in, n)) * fabsf(dot(out, n)));
      pdf = D * fabsf(dot(wh, n)) / (4.f * fabsf(dot(in, wh)));
      break;
    }
    case caster::MT_SUBSTRATE: {
      if (!SameHemiSphere(in, out, nor)) { fr = v3(0, 0, 0); pdf = 0; return; }
      V3 n = nor;
      if (dot(nor, in) < 0) n = -n;
      float cosThetaWo = fabsf(dot(in, n));
      float cosThetaWi = fabsf(dot(out, n));
      V3 Rd = GetTexel(cx, material, uv);
      V3 Rs = v3(material.specular);
      V3 diffuse = (28.f / (23.f * PI)) * Rd * (v3(1.f, 1.f, 1.f) - Rs) *
                   (1 - pow5(1 - .5f * cosThetaWi)) * (1 - pow5(1 - .5f * cosThetaWo));
      V3 wh = normalize(in + out);
      float D = GGX_D(wh, n, dpdu, material.alphaU, material.alphaV);
      float mx = cosThetaWi > cosThetaWo ? cosThetaWi : cosThetaWo;
      V3 specular = D / (4.f * fabsf(dot(out, wh)) * mx) * SchlickFresnel(Rs, dot(out, wh));
      fr = diffuse + specular;
      pdf = 0.5f * (fabsf(dot(out, n)) * ONE_OVER_PI + D * fabsf(dot(wh, n)) / (4.f * dot(in, wh)));
      break;
    }
    case caster::MT_ROUGHDIELECTRIC: {
      V3 wo = -in;
      V3 n = nor;
      bool reflect = dot(in, n) * dot(out, n) > 0;
      float ei = material.outsideIOR, et = material.insideIOR;
      float cosi = dot(wo, n);
      if (cosi > 0) { float t = et; et = ei; ei = t; }
      V3 wh = normalize(-(ei * in + et * out));
      float eta = ei / et;
      float D = GGX_D(wh, n, dpdu, material.alphaU, material.alphaV);
      float cosThetaI = dot(wo, wh);
      float sin2ThetaI = fps::fmax_(0.f, 1.f - cosThetaI * cosThetaI);
      float sin2ThetaT = eta * eta * sin2ThetaI;
      float cosThetaT = sqrtf(1.f - sin2ThetaT < 0.f ? 0.f : 1.f - sin2ThetaT);
      float fresnel = DielectricFresnel(fabsf(cosThetaI), fabsf(cosThetaT), ei, et);
      if (!reflect) {
        float G = GGX_G(in, out, n, wh, dpdu, material.alphaU, material.alphaV);
        float c = et * dot(out, wh) + ei * dot(in, wh);
        fr = v3(material.specular) * ei * ei * D * G * (1.f - fresnel) * fabsf(dot(in, wh)) *
             fabsf(dot(out, wh)) / (fabsf(dot(out, n)) * fabsf(dot(in, n)) * c * c);
        fr *= (1.f / (eta * eta));
        pdf = (1.f - fresnel) * D * fabsf(dot(wh, n)) * et * et * fabsf(dot(out, wh)) / (c * c);
      } else {
        float G = GGX_G(in, out, n, wh, dpdu, material.alphaU, material.alphaV);
        fr = v3(material.specular) * fresnel * D * G / (4.f * fabsf(dot(in, n)) * fabsf(dot(out, n)));
        pdf = fresnel * D * fabsf(dot(wh, n)) / (4.f * fabsf(dot(wh, in)));
      }
      break;
    }
  }
}

// ----------------------------------------------------------------- lights
inline float TriArea(const Prim& p) {  // Triangle.h:81-85
  V3 e1 = v3(p.v[1].v) - v3(p.v[0].v), e2 = v3(p.v[2].v) - v3(p.v[0].v);
  return length(cross(e1, e2)) * 0.5f;
}
// Triangle.h:146-155
inline void TriSampleShape(const Prim& p, V3 pos, V2 u, V3& dir, V3& nor, float& pdf) {
  V2 uv = UniformSampleTriangle(u.x, u.y);
  V3 pt = uv.x * v3(p.v[0].v) + uv.y * v3(p.v[1].v) + (1 - uv.x - uv.y) * v3(p.v[2].v);
  V3 normal = normalize(uv.x * v3(p.v[0].n) + uv.y * v3(p.v[1].n) + (1 - uv.x - uv.y) * v3(p.v[2].n));
  dir = pt - pos;
  nor = normal;
  pdf = 1.f / (TriArea(p) * fabsf(dot(normal, normalize(dir)))) * dot(dir, dir);
  if (dot(normal, dir) >= 0.f) pdf = 0.f;
}
// area.h:12-17
inline void AreaSampleLight(const Ctx& cx, int idx, V3 pos, V2 u, V3& rad, Ray& ray, V3& nor, float& pdf) {
  const caster::Light& L = cx.scene.lights[idx];
  V3 dir;
  TriSampleShape(cx.scene.prims[L.prim], pos, u, dir, nor, pdf);
  rad = pdf != 0.f ? v3(L.radiance) : v3(0.f, 0.f, 0.f);
  float eps = cx.scene.epsilon;
  ray = Ray(pos, normalize(dir), -1, eps, sqrtf(dot(dir, dir) - eps));
}
// area.h:26-30
inline void AreaPdf(const Ctx& cx, int idx, V3 raydir, V3 n, float& pdfA, float& pdfW) {
  float sa = TriArea(cx.scene.prims[cx.scene.lights[idx].prim]);
  pdfA = 1.f / sa;
  pdfW = fabsf(dot(raydir, n)) * ONE_OVER_PI;
}
// area.h:36-39
inline V3 AreaLe(const Ctx& cx, int idx, V3 n, V3 dir) {
  if (dot(n, dir) > 0.f) return v3(cx.scene.lights[idx].radiance);
  return v3(0, 0, 0);
}
// infinite.h:80-93
inline V3 InfTexel(const caster::Infinite& I, int x, int y) {
  int width = I.w, height = I.h;
  float rx = (float)(x - (x / width) * width);
  float ry = (float)(y - (y / height) * height);
  x = fps::f2i_((rx < 0) ? rx + width : rx);
  y = fps::f2i_((ry < 0) ? ry + height : ry);
  if (x < 0) x = 0;
  if (x > width - 1) x = width - 1;
  if (y < 0) y = 0;
  if (y > height - 1) y = height - 1;
  return v3(I.data[(size_t)y * width + x]);
}
// infinite.h:65-78
inline V3 InfBilinear(const caster::Infinite& I, V2 uv) {
  float xx = I.w * uv.x;
  float yy = I.h * uv.y;
  int x = fps::f2i_(floorf(xx));
  int y = fps::f2i_(floorf(yy));
  float dx = fabsf(xx - x);
  float dy = fabsf(yy - y);
  V3 c00 = InfTexel(I, x, y), c10 = InfTexel(I, x + 1, y);
  V3 c01 = InfTexel(I, x, y + 1), c11 = InfTexel(I, x + 1, y + 1);
  return (1 - dy) * ((1 - dx) * c00 + dx * c10) + dy * ((1 - dx) * c01 + dx * c11);
}
// infinite.h:46-58
inline V3 InfLe(const caster::Infinite& I, V3 dir) {
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
  return InfBilinear(I, V2{1.f - uu, vv});
}
// infinite.h:16-35
inline void InfSampleLight(const Ctx& cx, V3 pos, V2 un, V3& rad, Ray& ray, V3& nor, float& pdf) {
  const caster::Infinite& I = cx.scene.infinite;
  float pdfW;
  V3 dir = UniformSampleSphere(un.x, un.y, pdfW);
  nor = -dir;
  float eps = cx.scene.epsilon;
  ray = Ray(pos, dir, -1, eps, 2.f * I.radius - eps);
  pdf = pdfW;
  rad = InfLe(I, dir);  // same lookup as lines 21-34
}
// infinite.h:37-40
inline void InfPdf(const caster::Infinite& I, float& pdfA, float& pdfW) {
  pdfW = ONE_OVER_FOUR_PI;
  pdfA = 1.f / (PI * I.radius * I.radius);
}

// pathtracer.cu:181-190
inline int LookUpLightDistribution(const Ctx& cx, float u, float& pdf) {
  const std::vector<float>& d = cx.scene.light_cdf;
  int n = (int)d.size();
  for (int i = 0; i + 1 < n; ++i) {
    float s = d[i], e = d[i + 1];
    if (u >= s && u <= e) { pdf = e - s; return i; }
  }
  int last = n - 2;
  if (last < 0) { pdf = 0.f; return 0; }
  pdf = d[last + 1] - d[last];
  return last;
}
inline float PdfFromLightDistribution(const Ctx& cx, int idx) {  // :192-194
  return cx.scene.light_cdf[idx + 1] - cx.scene.light_cdf[idx];
}
inline bool HasLights(const Ctx& cx) { return !cx.scene.lights.empty() || cx.scene.infinite.valid; }

// Light sampling step shared by Path and Volpath (pathtracer.cu:921-930 /
// 1064-1074 / 1120-1129): three draws, then the chosen light's SampleLight.
inline void SampleOneLight(const Ctx& cx, Rng& rng, V3 pos, V3& radiance, Ray& shadowRay, float& lightPdf,
                           float& choicePdf) {
  int idx = LookUpLightDistribution(cx, rng.uniform(), choicePdf);
  V2 u1;
  u1.x = rng.uniform();
  u1.y = rng.uniform();
  V3 lightNor;
  if (!HasLights(cx)) {
    radiance = v3(0, 0, 0);
    lightPdf = 1.f;
    shadowRay = Ray(pos, v3(0, 0, 1), -1, cx.scene.epsilon, 0.f);
    return;
  }
  if (idx != (int)cx.scene.lights.size()) AreaSampleLight(cx, idx, pos, u1, radiance, shadowRay, lightNor, lightPdf);
  else InfSampleLight(cx, pos, u1, radiance, shadowRay, lightNor, lightPdf);
}

// ----------------------------------------------------------------- camera
// Camera.h:45-85
inline Ray GeneratePrimaryRay(const caster::Camera& c, float x, float y, V2 xy) {
  V3 cu = v3(c.u), cv = v3(c.v), cw = v3(c.w);
  if (c.environment) {
    V3 orig = v3(c.position);
    float theta = PI * (1.f - y / c.res_y);
    float phi = TWOPI * (1.f - x / c.res_x);
    V3 dir = v3(fps::sin_(theta) * fps::cos_(phi), fps::cos_(theta), fps::sin_(theta) * fps::sin_(phi));
    dir = dir.x * cu + dir.y * cv - dir.z * cw;
    return Ray(orig, dir);
  }
  float xx = x * c.p2s_x - c.half_w;
  float yy = y * c.p2s_y - c.half_h;
  V3 dir, orig = v3(c.position);
  if (c.apertureRadius > 0.00001f) {
    V2 aperture_xy = V2{xy.x * c.apertureRadius, xy.y * c.apertureRadius};
    float focal_x = c.ratio * xx;
    float focal_y = c.ratio * yy;
    V3 aperture = v3(aperture_xy.x, aperture_xy.y, 0);
    V3 focal = v3(focal_x, focal_y, -c.focalDistance);
    dir = focal - aperture;
    dir = dir.x * cu + dir.y * cv + dir.z * cw;
    orig += (aperture.x * cu + aperture.y * cv);
  } else {
    dir = xx * cu + yy * cv + -c.distance * cw;
  }
  dir = normalize(dir);
  Ray ray;
  ray.origin = orig;
  ray.dir = dir;
  return ray;
}

// Common prologue of the three kernels (pathtracer.cu:823-839 / 874-890 /
// 1016-1033): pixel = x + y*W, seed, four draws, primary ray.
inline Ray CameraSample(const Ctx& cx, uint32_t pixel, uint32_t iter, Rng& rng) {
  uint32_t W = (uint32_t)cx.scene.width;
  uint32_t x = pixel % W, y = pixel / W;
  float offsetx = rng.uniform() - 0.5f;
  float offsety = rng.uniform() - 0.5f;
  float a1 = rng.uniform();
  float a2 = rng.uniform();
  V2 aperture = UniformSampleDisk(a1, a2);
  Ray ray = GeneratePrimaryRay(cx.scene.camera, x + offsetx, y + offsety, aperture);
  ray.tmin = cx.scene.epsilon;
  return ray;
}
inline uint32_t SampleSeed(uint32_t pixel, uint32_t iter) { return WangHash(pixel) + WangHash(iter); }

// ------------------------------------------------------------ integrators
// pathtracer.cu:822-869
bool Ao(const Ctx& cx, uint32_t pixel, uint32_t iter, V3& color, Counters& k) {
  Rng rng(SampleSeed(pixel, iter));
  Ray ray = CameraSample(cx, pixel, iter, rng);
  k.c.samples++;
  V3 L = v3(0.f, 0.f, 0.f);
  Isect isect;
  if (!Intersect(cx, ray, &isect, k, RK_EXT)) { color = v3(1, 1, 1); return true; }
  k.c.bounces++;
  V3 pos = isect.pos;
  V3 n = isect.n;
  float pdf = 0.f;
  if (dot(-ray.dir, n) < 0.f) n = -n;
  float u1 = rng.uniform();
  float u2 = rng.uniform();
  V3 wi = CosineSampleHemiSphere(u1, u2, pdf);
  V3 s = isect.dpdu;
  V3 t = cross(n, s);
  wi = ToWorld(wi, s, t, n);
  float cosine = dot(wi, n);
  Ray r(pos, wi, -1, cx.scene.epsilon, cx.scene.maxDist);
  if (!IntersectPrimitive(cx, r, k, RK_SHADOW)) {
    float v = cosine / pdf;
    L += v3(v, v, v);
  }
  if (IsNan(L)) { k.c.invalid++; return false; }
  color = L;
  return true;
}

// Direct lighting by BSDF sampling (pathtracer.cu:943-983 and 1142-1194);
// `vol` adds the transmittance factor of the volumetric variant.
inline V3 BsdfSampledLight(const Ctx& cx, const Material& material, const Ray& r, V3 pos, V3 nor, V2 uv, V3 dpdu,
                           V3 us, bool vol, Rng& rng, Counters& k) {
  V3 Ld = v3(0, 0, 0);
  V3 out, fr;
  float pdf;
  SampleBSDF(cx, material, -r.dir, nor, uv, dpdu, us, out, fr, pdf);
  if (!IsBlack(fr) && pdf != 0) {
    Isect lightIsect;
    Ray lightRay(pos, out, r.medium, cx.scene.epsilon);
    if (Intersect(cx, lightRay, &lightIsect, k, RK_MIS)) {
      V3 p = lightIsect.pos;
      V3 n = lightIsect.n;
      V3 radiance = v3(0.f, 0.f, 0.f);
      if (lightIsect.lightIdx != -1) radiance = AreaLe(cx, lightIsect.lightIdx, n, -lightRay.dir);
      if (!IsBlack(radiance)) {
        float pdfA, pdfW;
        AreaPdf(cx, lightIsect.lightIdx, -out, n, pdfA, pdfW);
        float choicePdf = PdfFromLightDistribution(cx, lightIsect.lightIdx);
        float lenSquare = dot(p - pos, p - pos);
        float costheta = fabsf(dot(n, lightRay.dir));
        float lPdf = pdfA * lenSquare / (costheta);
        float weight = PowerHeuristic(1, pdf, 1, lPdf * choicePdf);
        if (vol) {
          V3 tr = v3(1.f, 1.f, 1.f);
          if (lightRay.medium != -1) tr = MediumTr(cx.scene.media[lightRay.medium], lightRay, rng);
          Ld += weight * tr * fr * radiance * fabsf(dot(out, nor)) / pdf;
        } else {
          Ld += weight * fr * radiance * fabsf(dot(out, nor)) / pdf;
        }
      }
    } else if (cx.scene.infinite.valid) {
      V3 radiance = InfLe(cx.scene.infinite, lightRay.dir);
      float choicePdf = PdfFromLightDistribution(cx, (int)cx.scene.lights.size());
      float lightPdf, pdfA;
      InfPdf(cx.scene.infinite, pdfA, lightPdf);
      float weight = PowerHeuristic(1, pdf, 1, lightPdf * choicePdf);
      if (vol) {
        V3 tr = v3(1.f, 1.f, 1.f);
        if (lightRay.medium != -1) tr = MediumTr(cx.scene.media[lightRay.medium], lightRay, rng);
        Ld += weight * tr * fr * radiance * fabsf(dot(out, nor)) / pdf;
      } else {
        Ld += weight * fr * radiance * fabsf(dot(out, nor)) / pdf;
      }
    }
  }
  return Ld;
}

// pathtracer.cu:873-1011
bool Path(const Ctx& cx, uint32_t pixel, uint32_t iter, V3& color, Counters& k) {
  Rng rng(SampleSeed(pixel, iter));
  Ray ray = CameraSample(cx, pixel, iter, rng);
  k.c.samples++;
  const int maxDepth = cx.scene.maxDepth;
  V3 Li = v3(0.f, 0.f, 0.f);
  V3 beta = v3(1.f, 1.f, 1.f);
  Ray r = ray;
  Isect isect;
  bool specular = false;
  for (int bounces = 0; bounces < maxDepth; ++bounces) {
    if (!Intersect(cx, r, &isect, k, RK_EXT)) {
      if ((bounces == 0 || specular) && cx.scene.infinite.valid) Li += beta * InfLe(cx.scene.infinite, r.dir);
      break;
    }
    k.c.bounces++;
    V3 pos = isect.pos;
    V3 nor = isect.n;
    V2 uv = isect.uv;
    V3 dpdu = isect.dpdu;
    const Material& material = cx.scene.materials[isect.matIdx];

    if (bounces == 0 || specular) {
      if (isect.lightIdx != -1) {
        Li += beta * AreaLe(cx, isect.lightIdx, nor, -r.dir);
        break;
      }
    }

    if (!IsDelta(material.type)) {
      V3 Ld = v3(0.f, 0.f, 0.f);
      float choicePdf, lightPdf;
      V3 radiance;
      Ray shadowRay;
      SampleOneLight(cx, rng, pos, radiance, shadowRay, lightPdf, choicePdf);
      shadowRay.medium = r.medium;
      if (!IsBlack(radiance) && !IntersectPrimitive(cx, shadowRay, k, RK_SHADOW)) {
        V3 fr;
        float samplePdf;
        Fr(cx, material, -r.dir, shadowRay.dir, nor, uv, dpdu, fr, samplePdf);
        float weight = PowerHeuristic(1, lightPdf * choicePdf, 1, samplePdf);
        Ld += weight * fr * radiance * fabsf(dot(nor, shadowRay.dir)) / (lightPdf * choicePdf);
      }
      V3 us;
      us.x = rng.uniform(); us.y = rng.uniform(); us.z = rng.uniform();
      Ld += BsdfSampledLight(cx, material, r, pos, nor, uv, dpdu, us, false, rng, k);
      Li += beta * Ld;
    }

    V3 u;
    u.x = rng.uniform(); u.y = rng.uniform(); u.z = rng.uniform();
    V3 out, fr;
    float pdf;
    SampleBSDF(cx, material, -r.dir, nor, uv, dpdu, u, out, fr, pdf);
    if (IsBlack(fr)) break;
    beta *= fr * fabsf(dot(nor, out)) / pdf;
    specular = IsDelta(material.type);
    r = Ray(pos, out, -1, cx.scene.epsilon);
    if (bounces > 3) {
      float illumate = fps::clamp_(1.f - Luminance(beta), 0.f, 1.f);
      if (rng.uniform() < illumate) break;
      beta /= (1 - illumate);
    }
  }
  if (IsInf(Li) || IsNan(Li)) { k.c.invalid++; return false; }
  color = Li;
  return true;
}

// pathtracer.cu:1015-1228
bool Volpath(const Ctx& cx, uint32_t pixel, uint32_t iter, V3& color, Counters& k) {
  Rng rng(SampleSeed(pixel, iter));
  Ray ray = CameraSample(cx, pixel, iter, rng);
  ray.medium = cx.scene.camera.medium;
  k.c.samples++;
  const int maxDepth = cx.scene.maxDepth;
  V3 Li = v3(0.f, 0.f, 0.f);
  V3 beta = v3(1.f, 1.f, 1.f);
  Ray r = ray;
  Isect isect;
  bool specular = false;
  int crossings = 0;
  for (int bounces = 0; bounces < maxDepth; ++bounces) {
    if (!Intersect(cx, r, &isect, k, RK_EXT)) {
      if ((bounces == 0 || specular) && cx.scene.infinite.valid) Li += beta * InfLe(cx.scene.infinite, r.dir);
      break;
    }
    k.c.bounces++;
    V3 pos = isect.pos;
    V3 nor = isect.n;
    V2 uv = isect.uv;
    V3 dpdu = isect.dpdu;

    float sampledDist = 0.f;
    bool sampledMedium = false;
    if (r.medium != -1) beta *= MediumSample(cx.scene.media[r.medium], r, rng, sampledDist, sampledMedium);
    if (IsBlack(beta)) break;
    if (sampledMedium) {
      const caster::Medium& med = cx.scene.media[r.medium];
      V3 samplePos = r.at(sampledDist);
      float choicePdf, lightPdf;
      V3 radiance;
      Ray shadowRay;
      // (the reference draws the light-pick number before forming samplePos;
      // neither consumes a draw out of order)
      SampleOneLight(cx, rng, samplePos, radiance, shadowRay, lightPdf, choicePdf);
      shadowRay.medium = r.medium;
      // The reference calls Tr() whatever the light sample is (pathtracer.cu:1079).
      // Tr() draws random numbers only inside a density grid, so without one in
      // the scene a black sample skips the walk; that changes the ray count only.
      V3 tr = (IsBlack(radiance) && !cx.scene.has_heterogeneous) ? v3(1, 1, 1) : Tr(cx, shadowRay, rng, k);
      float phase, unuse;
      Phase(med, -r.dir, shadowRay.dir, phase, unuse);
      if (!IsBlack(radiance)) Li += tr * beta * phase * radiance / (lightPdf * choicePdf);
      float pdf;
      V2 phaseU;
      phaseU.x = rng.uniform(); phaseU.y = rng.uniform();
      V3 dir;
      SamplePhase(med, phaseU, dir, phase, pdf);
      r = Ray(samplePos, dir, r.medium, cx.scene.epsilon);
      specular = false;
    } else {
      if (bounces == 0 || specular) {
        if (isect.lightIdx != -1) {
          V3 tr = v3(1.f, 1.f, 1.f);
          // (as the reference has it: beta already carries the transmittance of the medium sample that reached
          // this surface, so an emitter seen through a medium is attenuated twice -- tests/test_oracle_independent.py)
          if (r.medium != -1) tr = MediumTr(cx.scene.media[r.medium], r, rng);
          Li += tr * beta * AreaLe(cx, isect.lightIdx, nor, -r.dir);
          break;
        }
      }
      if (isect.matIdx == -1) {
        if (++crossings > kMaxCrossings) break;
        bounces--;
        int m = MediumBySide(dot(r.dir, isect.n), isect);
        r = Ray(pos, r.dir, m, cx.scene.epsilon);
        continue;
      }
      const Material& material = cx.scene.materials[isect.matIdx];
      if (!IsDelta(material.type)) {
        V3 Ld = v3(0.f, 0.f, 0.f);
        float choicePdf, lightPdf;
        V3 radiance;
        Ray shadowRay;
        SampleOneLight(cx, rng, pos, radiance, shadowRay, lightPdf, choicePdf);
        shadowRay.medium = r.medium;
        if (!IsBlack(radiance)) {
          V3 fr;
          float samplePdf;
          Fr(cx, material, -r.dir, shadowRay.dir, nor, uv, dpdu, fr, samplePdf);
          V3 tr = Tr(cx, shadowRay, rng, k);
          float weight = PowerHeuristic(1, lightPdf * choicePdf, 1, samplePdf);
          Ld += weight * tr * fr * radiance * fabsf(dot(nor, shadowRay.dir)) / (lightPdf * choicePdf);
        }
        V3 us;
        us.x = rng.uniform(); us.y = rng.uniform(); us.z = rng.uniform();
        Ld += BsdfSampledLight(cx, material, r, pos, nor, uv, dpdu, us, true, rng, k);
        Li += beta * Ld;
      }
      V3 u;
      u.x = rng.uniform(); u.y = rng.uniform(); u.z = rng.uniform();
      V3 out, fr;
      float pdf;
      SampleBSDF(cx, material, -r.dir, nor, uv, dpdu, u, out, fr, pdf);
      if (IsBlack(fr)) break;
      beta *= fr * fabsf(dot(nor, out)) / pdf;
      specular = IsDelta(material.type);
      int m = dot(out, nor) > 0 ? isect.mediumOutside : isect.mediumInside;
      m = dot(-r.dir, nor) * dot(out, nor) > 0 ? r.medium : m;
      r = Ray(pos, out, m, cx.scene.epsilon);
    }
    if (bounces > 3) {
      float illumate = fps::clamp_(1.f - Luminance(beta), 0.f, 1.f);
      if (rng.uniform() < illumate) break;
      beta /= (1 - illumate);
    }
  }
  if (IsInf(Li) || IsNan(Li)) { k.c.invalid++; return false; }
  color = Li;
  return true;
}

// Camera.h:87-118: connects `pos` to the pinhole.  pdf == 0: the camera does not see `pos`
// (we and raster are then left alone, as in the reference).
inline void SampleCamera(const Ctx& cx, V3 pos, Ray& ray, float& we, float& pdf, int& raster) {
  const caster::Camera& c = cx.scene.camera;
  V3 dir = v3(c.position) - pos;
  V3 ndir = normalize(dir);
  ray = Ray(pos, ndir, -1, cx.scene.epsilon, length(dir) - cx.scene.epsilon);
  V3 negative_dir = -ndir;
  V3 cndir = v3(dot(negative_dir, v3(c.u)), dot(negative_dir, v3(c.v)), dot(negative_dir, v3(c.w)));  // ToLocal, common.h:136-139
  if (cndir.z >= 0.f) { pdf = 0; return; }
  float costheta = dot(cndir, v3(0, 0, -1.f));
  float scale = -c.distance / cndir.z;
  cndir *= scale;
  V2 plane = V2{cndir.x / c.half_w, cndir.y / c.half_h};
  if (plane.x > 1.f || plane.x < -1.f || plane.y > 1.f || plane.y < -1.f) { pdf = 0; return; }
  plane = V2{plane.x * 0.5f + 0.5f, plane.y * 0.5f + 0.5f};
  int x = fps::f2i_(floorf(plane.x * (c.res_x - 1) + 0.5f));
  int y = fps::f2i_(floorf(plane.y * (c.res_y - 1) + 0.5f));
  raster = fps::f2i_(y * c.res_x + x);
  pdf = dot(dir, dir) / costheta;
  we = c.distance * c.distance / (c.area * costheta * costheta * costheta * costheta);
}

// Start of a light path (pathtracer.cu:1245-1261): seed, light choice, a point and a cosine
// distributed direction on the chosen triangle (area.h:19-24, Triangle.h:157-166).  False when
// no emissive triangle was chosen: the reference indexes kernel_lights with whatever
// LookUpLightDistribution returned, which is out of bounds for a scene without emitters and for
// the environment's bin; such a path draws its five numbers and carries nothing.
inline bool LightStart(const Ctx& cx, uint32_t pixel, uint32_t iter, Rng& rng, Ray& ray, V3& beta, V3& radiance) {
  beta = v3(1.f, 1.f, 1.f);
  float choicePdf;
  int lightIdx = LookUpLightDistribution(cx, rng.uniform(), choicePdf);
  V4 u;
  u.x = rng.uniform(); u.y = rng.uniform(); u.z = rng.uniform(); u.w = rng.uniform();
  if (lightIdx >= (int)cx.scene.lights.size()) return false;
  const caster::Light& light = cx.scene.lights[lightIdx];
  const Prim& p = cx.scene.prims[light.prim];
  V2 uv = UniformSampleTriangle(u.x, u.y);
  V3 pos = uv.x * v3(p.v[0].v) + uv.y * v3(p.v[1].v) + (1 - uv.x - uv.y) * v3(p.v[2].v);
  V3 nor = normalize(uv.x * v3(p.v[0].n) + uv.y * v3(p.v[1].n) + (1 - uv.x - uv.y) * v3(p.v[2].n));
  float pdfW;
  V3 dir = CosineSampleHemiSphere(u.z, u.w, pdfW);
  V3 uu, ww;
  MakeCoordinate(nor, uu, ww);
  dir = ToWorld(dir, uu, ww, nor);
  float pdfA = 1.f / TriArea(p);
  radiance = v3(light.radiance);
  ray = Ray(pos, dir, light.medium, cx.scene.epsilon);
  beta *= radiance * fabsf(dot(ray.dir, nor)) / (pdfA * pdfW * choicePdf);
  return true;
}

// pathtracer.cu:1240-1375.  `color` is the iteration's splat buffer (kernel_color, cleared by
// LightTracingInit :1232-1238); every splat is a plain add here, where the reference uses atomics
// for all but the first one (:1271, a data race between threads that share a raster position).
// A connection the camera cannot see (cameraPdf == 0) adds nothing: at :1297-1303 the reference
// divides uninitialised values by zero and discards the non-finite result.
void LightTracing(const Ctx& cx, uint32_t pixel, uint32_t iter, float* color, Counters& k) {
  Rng rng(SampleSeed(pixel, iter));
  k.c.samples++;
  const int maxDepth = cx.scene.maxDepth;
  V3 beta, radiance;
  Ray ray;
  if (!LightStart(cx, pixel, iter, rng, ray, beta, radiance)) return;
  auto splat = [&](int raster, V3 L) {
    if (raster < 0 || raster >= cx.scene.width * cx.scene.height) return;  // a NaN position
    color[3 * raster] += L.x; color[3 * raster + 1] += L.y; color[3 * raster + 2] += L.z;
  };
  {
    Ray shadowRay;
    float we = 0.f, cameraPdf = 0.f;
    int raster = 0;
    SampleCamera(cx, ray.origin, shadowRay, we, cameraPdf, raster);
    shadowRay.medium = ray.medium;
    if (cameraPdf != 0.f) {
      V3 tr = Tr(cx, shadowRay, rng, k);
      if (!IsBlack(tr)) splat(raster, tr * radiance);
    }
  }
  Isect isect;
  int crossings = 0;
  for (int bounces = 0; bounces < maxDepth; ++bounces) {
    if (!Intersect(cx, ray, &isect, k, RK_EXT)) break;
    k.c.bounces++;
    V3 pos = isect.pos;
    V3 nor = isect.n;
    V2 uv = isect.uv;
    float sampledDist = 0.f;
    bool sampledMedium = false;
    if (ray.medium != -1) beta *= MediumSample(cx.scene.media[ray.medium], ray, rng, sampledDist, sampledMedium);
    if (IsBlack(beta)) break;
    if (sampledMedium) {
      const caster::Medium& med = cx.scene.media[ray.medium];
      V3 samplePos = ray.at(sampledDist);
      Ray shadowRay;
      float we = 0.f, cameraPdf = 0.f;
      int raster = 0;
      SampleCamera(cx, samplePos, shadowRay, we, cameraPdf, raster);
      shadowRay.medium = ray.medium;
      // Tr() draws random numbers only inside a density grid: without one, a connection that
      // cannot contribute skips the walk (that changes the ray count only)
      if (cameraPdf != 0.f || cx.scene.has_heterogeneous) {
        V3 tr = Tr(cx, shadowRay, rng, k);
        float phase, unuse;
        Phase(med, -ray.dir, shadowRay.dir, phase, unuse);
        if (cameraPdf != 0.f) {
          V3 L = beta * we * tr * phase / cameraPdf;
          if (!IsInf(L) && !IsNan(L)) splat(raster, L);
        }
      }
      float phase, pdf;
      V2 phaseU;
      phaseU.x = rng.uniform(); phaseU.y = rng.uniform();
      V3 dir;
      SamplePhase(med, phaseU, dir, phase, pdf);
      ray = Ray(samplePos, dir, ray.medium, cx.scene.epsilon);
    } else {
      if (isect.matIdx == -1) {
        if (++crossings > kMaxCrossings) break;
        bounces--;
        int m = MediumBySide(dot(ray.dir, isect.n), isect);
        ray = Ray(pos, ray.dir, m, cx.scene.epsilon);
        continue;
      }
      const Material& mat = cx.scene.materials[isect.matIdx];
      if (!IsDelta(mat.type)) {
        Ray shadowRay;
        float we = 0.f, cameraPdf = 0.f;
        int raster = 0;
        SampleCamera(cx, pos, shadowRay, we, cameraPdf, raster);
        shadowRay.medium = ray.medium;
        if (cameraPdf != 0.f) {
          V3 tr = Tr(cx, shadowRay, rng, k);
          V3 fr;
          float unuse;
          Fr(cx, mat, -ray.dir, shadowRay.dir, nor, uv, isect.dpdu, fr, unuse);
          V3 L = tr * beta * fr * we * fabsf(dot(shadowRay.dir, nor)) / cameraPdf;
          if (!IsInf(L) && !IsNan(L)) splat(raster, L);
        }
      }
      V3 u;
      u.x = rng.uniform(); u.y = rng.uniform(); u.z = rng.uniform();
      V3 out, fr;
      float pdf;
      SampleBSDF(cx, mat, -ray.dir, nor, uv, isect.dpdu, u, out, fr, pdf, false);
      if (IsBlack(fr)) break;
      beta *= fr * fabsf(dot(out, nor)) / pdf;
      int m = dot(out, nor) > 0 ? isect.mediumOutside : isect.mediumInside;
      m = dot(-ray.dir, nor) * dot(out, nor) > 0 ? ray.medium : m;
      ray = Ray(pos, out, m, cx.scene.epsilon);
    }
    if (bounces > 3) {
      float illumate = fps::clamp_(1.f - Luminance(beta), 0.f, 1.f);
      if (rng.uniform() < illumate) break;
      beta /= (1 - illumate);
    }
  }
}

// ------------------------------------------------------------------- BDPT
// pathtracer.cu:1379-1956.  Conventions fixed here (mirrored by the HIP renderer):
//   * fwd/rev of a vertex are read by MisWeight only after they were written (checked below
//     for every (s, t)); they start at zero instead of the reference's uninitialised memory;
//   * a light choice that is not an emissive triangle (no emitters, or the environment's bin,
//     where the reference reads past kernel_lights) gives an empty light path, and a t == 1
//     connection to such a choice draws its three numbers and adds nothing;
//   * at most kMaxCrossings index-matched boundary crossings per sub-path;
//   * maxDepth is not used (the reference's kernel ignores it, :1919-1956): sub-paths end by
//     Russian roulette or after BDPT_MAX_DEPTH bounces.
const int kBdptMaxDepth = 65;  // BDPT_MAX_DEPTH, :1379

struct BdptVertex {  // :1381-1388
  V3 beta{0, 0, 0};
  Isect isect;
  int medium = -1;
  bool delta = false;
  float fwd = 0.f, rev = 0.f;
};

// Camera.h:121-126
inline void PdfCamera(const caster::Camera& c, V3 dir, float& pdfA, float& pdfW) {
  pdfA = 1.f;
  float costheta = dot(dir, -v3(c.w));
  pdfW = c.distance * c.distance / (c.area * costheta * costheta * costheta);
}

// :1391-1399
inline float ConvertPdf(float pdf, const Isect& prev, const Isect& cur) {
  V3 dir = prev.pos - cur.pos;
  float square = dot(dir, dir);
  dir = normalize(dir);
  float ret = pdf / square;
  if (!IsBlack(cur.n)) ret *= fabsf(dot(dir, cur.n));
  return ret;
}

// The walk shared by GenerateCameraPath (:1431-1534) and GenerateLightPath (:1569-1671): the two
// loops differ in the transport mode of SampleBSDF only.  path[0] is set by the caller.
int BdptWalk(const Ctx& cx, Ray ray, V3 beta, float forward, bool radiance, BdptVertex* path, Rng& rng, Counters& k) {
  int nVertex = 1;
  Isect isect;
  int crossings = 0;
  for (int bounces = 0; bounces < kBdptMaxDepth; ++bounces) {
    if (!Intersect(cx, ray, &isect, k, RK_EXT)) break;
    k.c.bounces++;
    V3 pos = isect.pos;
    V3 nor = isect.n;
    V2 uv = isect.uv;
    float sampledDist = 0.f;
    bool sampledMedium = false;
    if (ray.medium != -1) beta *= MediumSample(cx.scene.media[ray.medium], ray, rng, sampledDist, sampledMedium);
    if (IsBlack(beta)) break;
    if (sampledMedium) {
      const caster::Medium& med = cx.scene.media[ray.medium];
      V3 samplePos = ray.at(sampledDist);
      float phase, pdf;
      V2 phaseU;
      phaseU.x = rng.uniform(); phaseU.y = rng.uniform();
      V3 dir;
      SamplePhase(med, phaseU, dir, phase, pdf);
      ray = Ray(samplePos, dir, ray.medium, cx.scene.epsilon);
      BdptVertex vertex;
      vertex.isect.pos = samplePos;
      vertex.isect.n = v3(0.f, 0.f, 0.f);
      vertex.isect.matIdx = -1;
      vertex.isect.lightIdx = -1;
      vertex.beta = beta;
      vertex.delta = false;
      vertex.medium = ray.medium;
      path[bounces + 1] = vertex;
      path[bounces + 1].fwd = ConvertPdf(forward, path[bounces].isect, path[bounces + 1].isect);
      forward = phase;
      path[bounces].rev = ConvertPdf(forward, path[bounces + 1].isect, path[bounces].isect);
      nVertex++;
    } else {
      if (isect.matIdx == -1) {
        if (++crossings > kMaxCrossings) break;
        bounces--;
        int m = MediumBySide(dot(ray.dir, isect.n), isect);
        ray = Ray(pos, ray.dir, m, cx.scene.epsilon);
        continue;
      }
      const Material& mat = cx.scene.materials[isect.matIdx];
      {
        BdptVertex vertex;
        vertex.beta = beta;
        vertex.isect = isect;
        vertex.delta = IsDelta(mat.type);
        vertex.medium = ray.medium;
        path[bounces + 1] = vertex;
        path[bounces + 1].fwd = ConvertPdf(forward, path[bounces].isect, path[bounces + 1].isect);
      }
      nVertex++;
      V3 u;
      u.x = rng.uniform(); u.y = rng.uniform(); u.z = rng.uniform();
      V3 out, fr;
      float pdf;
      SampleBSDF(cx, mat, -ray.dir, nor, uv, isect.dpdu, u, out, fr, pdf, radiance);
      if (IsBlack(fr)) break;
      beta *= fr * fabsf(dot(out, nor)) / pdf;
      forward = pdf;
      if (IsDelta(mat.type)) forward = 0.f;
      {
        V3 unuseFr;
        float rpdf;
        Fr(cx, mat, out, -ray.dir, nor, uv, isect.dpdu, unuseFr, rpdf);
        path[bounces].rev = ConvertPdf(rpdf, path[bounces + 1].isect, path[bounces].isect);
      }
      int m = dot(out, nor) > 0 ? isect.mediumOutside : isect.mediumInside;
      m = dot(-ray.dir, nor) * dot(out, nor) > 0 ? ray.medium : m;
      ray = Ray(pos, out, m, cx.scene.epsilon);
    }
    if (bounces > 3) {
      float rrPdf = fps::clamp_(1.f - Luminance(beta), 0.f, 1.f);
      if (rng.uniform() < rrPdf) break;
      beta /= (1 - rrPdf);
    }
  }
  return nVertex;
}

// The camera ray of a BDPT sample (:1403-1410): two jitter draws, no lens sample.
inline Ray BdptCameraRay(const Ctx& cx, uint32_t pixel, Rng& rng) {
  uint32_t W = (uint32_t)cx.scene.width;
  uint32_t x = pixel % W, y = pixel / W;
  float offsetx = rng.uniform() - 0.5f;
  float offsety = rng.uniform() - 0.5f;
  Ray ray = GeneratePrimaryRay(cx.scene.camera, x + offsetx, y + offsety, V2{0.f, 0.f});
  ray.tmin = cx.scene.epsilon;
  ray.medium = cx.scene.camera.medium;
  return ray;
}

// :1401-1537
int GenerateCameraPath(const Ctx& cx, uint32_t pixel, BdptVertex* path, Rng& rng, Counters& k) {
  Ray ray = BdptCameraRay(cx, pixel, rng);
  BdptVertex vertex;
  vertex.isect.pos = v3(cx.scene.camera.position);
  vertex.isect.n = -v3(cx.scene.camera.w);
  vertex.beta = v3(1.f, 1.f, 1.f);
  vertex.delta = false;
  vertex.medium = ray.medium;
  vertex.fwd = 1.f;
  path[0] = vertex;
  float forward = 0.f, unuse;
  PdfCamera(cx.scene.camera, ray.dir, unuse, forward);
  return BdptWalk(cx, ray, v3(1.f, 1.f, 1.f), forward, true, path, rng, k);
}

// :1539-1674
int GenerateLightPath(const Ctx& cx, BdptVertex* path, Rng& rng, Counters& k) {
  float choicePdf;
  int lightIdx = LookUpLightDistribution(cx, rng.uniform(), choicePdf);
  V4 u;
  u.x = rng.uniform(); u.y = rng.uniform(); u.z = rng.uniform(); u.w = rng.uniform();
  if (lightIdx >= (int)cx.scene.lights.size()) return 0;
  const caster::Light& light = cx.scene.lights[lightIdx];
  const Prim& p = cx.scene.prims[light.prim];
  // Area::SampleLight(u, ...) (area.h:19-24) = Triangle::SampleShape (Triangle.h:157-166)
  V2 uv = UniformSampleTriangle(u.x, u.y);
  V3 pos = uv.x * v3(p.v[0].v) + uv.y * v3(p.v[1].v) + (1 - uv.x - uv.y) * v3(p.v[2].v);
  V3 lightNor = normalize(uv.x * v3(p.v[0].n) + uv.y * v3(p.v[1].n) + (1 - uv.x - uv.y) * v3(p.v[2].n));
  float pdfW;
  V3 dir = CosineSampleHemiSphere(u.z, u.w, pdfW);
  V3 uu, ww;
  MakeCoordinate(lightNor, uu, ww);
  dir = ToWorld(dir, uu, ww, lightNor);
  float pdfA = 1.f / TriArea(p);
  V3 radiance = v3(light.radiance);
  Ray ray(pos, dir, light.medium, cx.scene.epsilon);
  BdptVertex vertex;
  vertex.isect.pos = ray.origin;
  vertex.isect.n = lightNor;
  vertex.isect.lightIdx = lightIdx;
  vertex.beta = radiance;
  vertex.delta = false;
  vertex.medium = ray.medium;
  vertex.fwd = pdfA * choicePdf;
  path[0] = vertex;
  V3 beta = v3(1.f, 1.f, 1.f);
  beta *= radiance * fabsf(dot(ray.dir, lightNor)) / (pdfA * pdfW * choicePdf);
  return BdptWalk(cx, ray, beta, pdfW, false, path, rng, k);
}

// :1676-1704
float MisWeight(const BdptVertex* cameraPath, const BdptVertex* lightPath, int s, int t) {
  if (s + t == 2) return 1.f;
  auto remap = [](float pdf) -> float { return pdf == 0 ? 1.f : pdf; };
  float sumW = 0.f;
  float ri = 1.f;
  for (int i = s - 1; i > 0; --i) {
    ri *= remap(cameraPath[i].rev) / remap(cameraPath[i].fwd);
    if (!cameraPath[i].delta && !cameraPath[i - 1].delta) sumW += ri;
  }
  ri = 1.f;
  for (int i = t - 1; i >= 0; --i) {
    ri *= remap(lightPath[i].rev) / remap(lightPath[i].fwd);
    bool delta = lightPath[i == 0 ? 0 : i - 1].delta;
    if (!lightPath[i].delta && !delta) sumW += ri;
  }
  return 1.f / (sumW + 1.f);
}

// Phase function or BSDF of a sub-path vertex, as Connect() picks between them.
inline void VertexFr(const Ctx& cx, bool useMedium, const BdptVertex& v, const Material* mat, V3 in, V3 out, V3& fr,
                     float& pdf) {
  if (useMedium) {
    float phase;
    Phase(cx.scene.media[v.medium], in, out, phase, pdf);
    fr = v3(phase, phase, phase);
  } else {
    Fr(cx, *mat, in, out, v.isect.n, v.isect.uv, v.isect.dpdu, fr, pdf);
  }
}

// :1706-1909
V3 Connect(const Ctx& cx, BdptVertex* cameraPath, BdptVertex* lightPath, int s, int t, int& raster, Rng& rng,
           Counters& k) {
  const V3 zero = v3(0.f, 0.f, 0.f);
  V3 L = zero;
  const float eps = cx.scene.epsilon;
  if (t == 0) {  // the camera path hit an emitter
    BdptVertex& cur = cameraPath[s - 1];
    BdptVertex& prev = cameraPath[s - 2];
    if (cur.isect.lightIdx == -1) return zero;
    V3 dir = normalize(prev.isect.pos - cur.isect.pos);
    L += cur.beta * AreaLe(cx, cur.isect.lightIdx, cur.isect.n, dir);
    if (IsBlack(L)) return L;
    float choicePdf = PdfFromLightDistribution(cx, cur.isect.lightIdx);
    float pdfA, pdfW;
    AreaPdf(cx, cur.isect.lightIdx, dir, cur.isect.n, pdfA, pdfW);
    float curRev = cur.rev, prevRev = prev.rev;
    cur.rev = pdfA * choicePdf;
    prev.rev = ConvertPdf(pdfW, cur.isect, prev.isect);
    float mis = MisWeight(cameraPath, lightPath, s, t);
    cur.rev = curRev;
    prev.rev = prevRev;
    return mis * L;
  } else if (t == 1) {  // next event estimation from the camera path
    BdptVertex& prev = cameraPath[s - 2];
    BdptVertex& cur = cameraPath[s - 1];
    BdptVertex& next = lightPath[0];
    V3 in = normalize(prev.isect.pos - cur.isect.pos);
    bool isMedium = cur.isect.matIdx == -1;
    const Material* mat = isMedium ? nullptr : &cx.scene.materials[cur.isect.matIdx];
    float choicePdf, lightPdf;
    int idx = LookUpLightDistribution(cx, rng.uniform(), choicePdf);
    V2 lightUniform;
    lightUniform.x = rng.uniform(); lightUniform.y = rng.uniform();
    if (idx >= (int)cx.scene.lights.size()) return zero;
    V3 radiance, lightNor, lightPos;
    Ray shadowRay;
    AreaSampleLight(cx, idx, cur.isect.pos, lightUniform, radiance, shadowRay, lightNor, lightPdf);
    lightPos = shadowRay.at(shadowRay.tmax + eps);
    shadowRay.medium = cur.medium;
    if (IsBlack(radiance)) return zero;
    if (!isMedium && IsDelta(mat->type)) return zero;
    V3 sdir = shadowRay.dir;
    V3 tr = Tr(cx, shadowRay, rng, k);
    if (IsBlack(tr)) return zero;
    V3 fr;
    float nextPdf, G;
    VertexFr(cx, isMedium, cur, mat, in, sdir, fr, nextPdf);
    G = isMedium ? 1.f : fabsf(dot(cur.isect.n, sdir));
    L += cur.beta * tr * fr * radiance * G / (lightPdf * choicePdf);
    if (IsBlack(L)) return zero;
    BdptVertex tNext = next;
    float pdfA, pdfW;
    AreaPdf(cx, idx, sdir, lightNor, pdfA, pdfW);
    next.isect.pos = lightPos;
    next.isect.n = lightNor;
    next.fwd = pdfA * choicePdf;
    next.rev = ConvertPdf(nextPdf, cur.isect, next.isect);
    float curRev = cur.rev, prevRev = prev.rev;
    cur.rev = ConvertPdf(pdfW, next.isect, cur.isect);
    float pdf;
    VertexFr(cx, isMedium, cur, mat, sdir, in, fr, pdf);
    prev.rev = ConvertPdf(pdf, cur.isect, prev.isect);
    float mis = MisWeight(cameraPath, lightPath, s, t);
    cur.rev = curRev;
    prev.rev = prevRev;
    next = tNext;
    return mis * L;
  } else if (s == 1) {  // the light path connected to the pinhole
    BdptVertex& prev = lightPath[t - 2];
    BdptVertex& cur = lightPath[t - 1];
    BdptVertex& next = cameraPath[0];
    V3 in = normalize(prev.isect.pos - cur.isect.pos);
    bool isMedium = cur.isect.matIdx == -1;
    const Material* mat = isMedium ? nullptr : &cx.scene.materials[cur.isect.matIdx];
    Ray shadowRay;
    float we = 0.f, cameraPdf = 0.f;
    SampleCamera(cx, cur.isect.pos, shadowRay, we, cameraPdf, raster);
    shadowRay.medium = cur.medium;
    if (cameraPdf == 0) return zero;
    if (!isMedium && IsDelta(mat->type)) return zero;
    V3 sdir = shadowRay.dir;
    V3 tr = Tr(cx, shadowRay, rng, k);
    if (IsBlack(tr)) return zero;
    V3 fr;
    float nextPdf, costheta = fabsf(dot(sdir, cur.isect.n));
    VertexFr(cx, isMedium, cur, mat, in, sdir, fr, nextPdf);
    if (isMedium) costheta = 1.f;
    L += cur.beta * tr * fr * we * costheta / cameraPdf;
    if (IsBlack(L)) return zero;
    float nextRev = next.rev, curRev = cur.rev, prevRev = prev.rev;
    next.rev = ConvertPdf(nextPdf, cur.isect, next.isect);
    float pdfA, pdfW;
    PdfCamera(cx.scene.camera, -sdir, pdfA, pdfW);
    cur.rev = ConvertPdf(pdfW, next.isect, cur.isect);
    float pdf;
    VertexFr(cx, isMedium, cur, mat, sdir, in, fr, pdf);
    prev.rev = ConvertPdf(pdf, cur.isect, prev.isect);
    float mis = MisWeight(cameraPath, lightPath, s, t);
    next.rev = nextRev;
    cur.rev = curRev;
    prev.rev = prevRev;
    return mis * L;
  }
  // an inner vertex of each path joined by a shadow ray.  Here the reference tells a medium
  // vertex from a surface vertex by the vertex's medium, not by its material (:1854-1881): a
  // surface reached through a medium is given that medium's phase function.
  BdptVertex& c2 = cameraPath[s - 2];
  BdptVertex& c1 = cameraPath[s - 1];
  BdptVertex& l1 = lightPath[t - 1];
  BdptVertex& l2 = lightPath[t - 2];
  V3 l1Tol2 = normalize(l2.isect.pos - l1.isect.pos);
  V3 l1Toc1 = normalize(c1.isect.pos - l1.isect.pos);
  V3 c1Tol1 = -l1Toc1;
  V3 c1Toc2 = normalize(c2.isect.pos - c1.isect.pos);
  V3 dir = c1.isect.pos - l1.isect.pos;
  bool c1Med = c1.medium != -1, l1Med = l1.medium != -1;
  const Material* c1Mat = c1Med ? nullptr : &cx.scene.materials[c1.isect.matIdx];
  const Material* l1Mat = l1Med ? nullptr : &cx.scene.materials[l1.isect.matIdx];
  Ray shadowRay;
  shadowRay.origin = c1.isect.pos;
  shadowRay.dir = c1Tol1;
  shadowRay.medium = c1.medium;
  shadowRay.tmin = eps;
  shadowRay.tmax = length(dir) - eps;
  if (!c1Med && IsDelta(c1Mat->type)) return zero;
  if (!l1Med && IsDelta(l1Mat->type)) return zero;
  V3 tr = Tr(cx, shadowRay, rng, k);
  if (IsBlack(tr)) return zero;
  float cos1 = l1Med ? 1.f : fabsf(dot(l1Toc1, l1.isect.n));
  float cos2 = c1Med ? 1.f : fabsf(dot(c1Tol1, c1.isect.n));
  V3 c1Fr, l1Fr;
  float l1Pdf, c1Pdf;
  VertexFr(cx, c1Med, c1, c1Mat, c1Toc2, c1Tol1, c1Fr, l1Pdf);
  VertexFr(cx, l1Med, l1, l1Mat, l1Tol2, l1Toc1, l1Fr, c1Pdf);
  V3 G = tr * cos1 * cos2 / dot(dir, dir);
  L += c1.beta * c1Fr * G * l1Fr * l1.beta;
  if (IsBlack(L)) return zero;
  float c2Rev = c2.rev, c1Rev = c1.rev, l1Rev = l1.rev, l2Rev = l2.rev;
  c1.rev = ConvertPdf(c1Pdf, l1.isect, c1.isect);
  l1.rev = ConvertPdf(l1Pdf, c1.isect, l1.isect);
  float l2Pdf, c2Pdf;
  VertexFr(cx, l1Med, l1, l1Mat, l1Toc1, l1Tol2, l1Fr, l2Pdf);
  VertexFr(cx, c1Med, c1, c1Mat, c1Tol1, c1Toc2, c1Fr, c2Pdf);
  l2.rev = ConvertPdf(l2Pdf, l1.isect, l2.isect);
  c2.rev = ConvertPdf(c2Pdf, c1.isect, c2.isect);
  float mis = MisWeight(cameraPath, lightPath, s, t);
  c2.rev = c2Rev;
  c1.rev = c1Rev;
  l1.rev = l1Rev;
  l2.rev = l2Rev;
  return mis * L;
}

// :1919-1956.  `color` is the iteration's buffer (kernel_color, cleared by BdptInit :1911-1917);
// the reference's atomic adds are plain adds here.
void Bdpt(const Ctx& cx, uint32_t pixel, uint32_t iter, float* color, Counters& k) {
  Rng rng(SampleSeed(pixel, iter));
  k.c.samples++;
  BdptVertex cameraPath[kBdptMaxDepth + 2];
  BdptVertex lightPath[kBdptMaxDepth + 2];
  int nCamera = GenerateCameraPath(cx, pixel, cameraPath, rng, k);
  int nLight = GenerateLightPath(cx, lightPath, rng, k);
  const int nPix = cx.scene.width * cx.scene.height;
  for (int s = 1; s <= nCamera; ++s) {
    for (int t = 0; t <= nLight; ++t) {
      if ((s == 1 && t == 0) || (s == 1 && t == 1)) continue;
      int raster = -1;
      V3 L = Connect(cx, cameraPath, lightPath, s, t, raster, rng, k);
      if (IsInf(L) || IsNan(L) || IsBlack(L)) continue;
      int where = s == 1 ? raster : (int)pixel;
      if (where < 0 || where >= nPix) continue;  // a NaN position
      color[3 * where] += L.x; color[3 * where + 1] += L.y; color[3 * where + 2] += L.z;
    }
  }
}

inline bool Sample(const Ctx& cx, uint32_t pixel, uint32_t iter, V3& color, Counters& k) {
  switch (cx.scene.integrator) {
    case caster::IT_AO: return Ao(cx, pixel, iter, color, k);
    case caster::IT_VPT: return Volpath(cx, pixel, iter, color, k);
    default: return Path(cx, pixel, iter, color, k);
  }
}

void add_counts(oracle_counts* dst, const oracle_counts& s) {
  for (int i = 0; i < 4; ++i) {
    dst->rays[i] += s.rays[i]; dst->nodes[i] += s.nodes[i]; dst->tris[i] += s.tris[i];
    dst->spheres[i] += s.spheres[i]; dst->lines[i] += s.lines[i]; dst->hits[i] += s.hits[i];
  }
  dst->samples += s.samples; dst->invalid += s.invalid; dst->bounces += s.bounces;
  dst->sphere_box_rejects += s.sphere_box_rejects;
}

}  // namespace

struct oracle_ctx { Ctx cx; };

extern "C" {

const char* oracle_last_error(void) { return g_err.c_str(); }

int oracle_load(const char* scene_json, oracle_ctx** out) {
  oracle_ctx* c = new oracle_ctx();
  std::string err;
  if (!caster::load_scene(scene_json, c->cx.scene, err)) {
    g_err = err;
    delete c;
    return 1;
  }
  c->cx.g1 = gamma_(1); c->cx.g2 = gamma_(2); c->cx.g3 = gamma_(3);
  *out = c;
  return 0;
}
void oracle_free(oracle_ctx* c) { delete c; }

void oracle_info(const oracle_ctx* c, int* o) {
  const Scene& s = c->cx.scene;
  o[0] = s.width; o[1] = s.height; o[2] = s.integrator; o[3] = (int)s.prims.size();
  o[4] = (int)s.nodes.size(); o[5] = (int)s.lights.size(); o[6] = (int)s.materials.size();
  o[7] = s.maxDepth;
}

int oracle_texture(const oracle_ctx* c, int idx, int* wh, unsigned char* rgba) {
  const Scene& s = c->cx.scene;
  if (idx < 0 || idx >= (int)s.textures.size()) return 1;
  wh[0] = s.textures[(size_t)idx].w; wh[1] = s.textures[(size_t)idx].h;
  if (rgba) memcpy(rgba, s.textures[(size_t)idx].rgba.data(), s.textures[(size_t)idx].rgba.size());
  return 0;
}

int oracle_render(oracle_ctx* c, uint32_t first_iter, uint32_t n_iter, int threads, float* acc,
                  oracle_counts* counts) {
  return oracle_render_rows(c, first_iter, n_iter, threads, 0, c->cx.scene.height, acc, counts);
}

int oracle_render_rows(oracle_ctx* c, uint32_t first_iter, uint32_t n_iter, int threads, int y0, int y1, float* acc,
                       oracle_counts* counts) {
  const Ctx& cx = c->cx;
  const uint32_t W = cx.scene.width;
  if (y0 < 0 || y1 > cx.scene.height || y0 > y1) return 1;
  const uint32_t H = (uint32_t)y1;
  if (threads <= 0) threads = 1;
  std::vector<Counters> ks(threads);
  if (cx.scene.integrator == caster::IT_LT || cx.scene.integrator == caster::IT_BDPT) {
    const bool bdpt = cx.scene.integrator == caster::IT_BDPT;
    // Light paths splat anywhere on the film: rows [y0, y1) select which paths are traced, not
    // where they land.  Each thread takes a fixed share of the rows and splats into a buffer of
    // its own; per iteration the buffers are summed in thread order (kernel_color) and added to
    // acc as Output() does (pathtracer.cu:2502-2513), so the result depends on `threads` only
    // in the rounding of those sums.
    const size_t n3 = (size_t)W * cx.scene.height * 3;
    std::vector<std::vector<float>> color(threads, std::vector<float>(n3));
    for (uint32_t it = 0; it < n_iter; ++it) {
      auto work = [&](int tid) {
        std::fill(color[tid].begin(), color[tid].end(), 0.f);
        uint32_t rows = H - (uint32_t)y0;
        uint32_t r0 = (uint32_t)y0 + (uint32_t)((uint64_t)rows * tid / threads);
        uint32_t r1 = (uint32_t)y0 + (uint32_t)((uint64_t)rows * (tid + 1) / threads);
        for (uint32_t y = r0; y < r1; ++y)
          for (uint32_t x = 0; x < W; ++x) {
            if (bdpt) Bdpt(cx, x + y * W, first_iter + it, color[tid].data(), ks[tid]);
            else LightTracing(cx, x + y * W, first_iter + it, color[tid].data(), ks[tid]);
          }
      };
      std::vector<std::thread> pool;
      for (int t = 1; t < threads; ++t) pool.emplace_back(work, t);
      work(0);
      for (auto& t : pool) t.join();
      for (size_t i = 0; i < n3; ++i) {
        float c = color[0][i];
        for (int t = 1; t < threads; ++t) c += color[t][i];
        acc[i] += c;
      }
    }
    if (counts) for (auto& k : ks) add_counts(counts, k.c);
    return 0;
  }
  std::atomic<uint32_t> next_row{(uint32_t)y0};
  auto work = [&](int tid) {
    Counters& k = ks[tid];
    for (;;) {
      uint32_t y = next_row.fetch_add(1);
      if (y >= H) break;
      for (uint32_t x = 0; x < W; ++x) {
        uint32_t pixel = x + y * W;
        for (uint32_t it = 0; it < n_iter; ++it) {
          V3 col;
          if (Sample(cx, pixel, first_iter + it, col, k)) {
            acc[3 * pixel] += col.x; acc[3 * pixel + 1] += col.y; acc[3 * pixel + 2] += col.z;
          }
        }
      }
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < threads; ++t) pool.emplace_back(work, t);
  work(0);
  for (auto& t : pool) t.join();
  if (counts) for (auto& k : ks) add_counts(counts, k.c);
  return 0;
}

void oracle_tonemap(const float* in, float* out, int n, int filmic) {
  for (int i = 0; i < n; ++i) {
    V3 c = v3(in[3 * i], in[3 * i + 1], in[3 * i + 2]);
    if (filmic) {  // pathtracer.cu:208-213
      c = c - v3(0.004f, 0.004f, 0.004f);
      c = v3(fps::fmax_(0.f, c.x), fps::fmax_(0.f, c.y), fps::fmax_(0.f, c.z));
      c = (c * (6.2f * c + 0.5f)) / (c * (6.2f * c + 1.7f) + 0.06f);
    } else {  // pathtracer.cu:196-206
      float g = 1.f / 2.2f, e = 1.41421356f;
      c = v3(fps::fmax_(c.x, 1e-5f), fps::fmax_(c.y, 1e-5f), fps::fmax_(c.z, 1e-5f));
      c = v3(fps::pow_(c.x * e, g), fps::pow_(c.y * e, g), fps::pow_(c.z * e, g));
    }
    out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
  }
}

int oracle_primary_ids(oracle_ctx* c, uint32_t iter, int32_t* ids) {
  const Ctx& cx = c->cx;
  Counters k;
  uint32_t n = (uint32_t)cx.scene.width * cx.scene.height;
  for (uint32_t p = 0; p < n; ++p) {
    Rng rng(SampleSeed(p, iter));
    Isect is;
    if (cx.scene.integrator == caster::IT_LT) {  // the first surface the light path of (p, iter) reaches
      V3 beta, radiance;
      Ray ray;
      ids[p] = LightStart(cx, p, iter, rng, ray, beta, radiance) && Intersect(cx, ray, &is, k, RK_EXT) ? is.prim : -1;
      continue;
    }
    Ray ray = cx.scene.integrator == caster::IT_BDPT ? BdptCameraRay(cx, p, rng) : CameraSample(cx, p, iter, rng);
    if (cx.scene.integrator == caster::IT_VPT) ray.medium = cx.scene.camera.medium;
    ids[p] = Intersect(cx, ray, &is, k, RK_EXT) ? is.prim : -1;
  }
  return 0;
}

int oracle_primary_ids_bruteforce(oracle_ctx* c, uint32_t iter, int32_t* ids) {
  const Ctx& cx = c->cx;
  Counters k;
  uint32_t n = (uint32_t)cx.scene.width * cx.scene.height;
  for (uint32_t p = 0; p < n; ++p) {
    Rng rng(SampleSeed(p, iter));
    Ray ray = CameraSample(cx, p, iter, rng);
    Isect is;
    bool hit = false;
    for (int i = 0; i < (int)cx.scene.prims.size(); ++i)
      if (PrimIntersect(cx, i, ray, &is, k, RK_EXT)) hit = true;
    ids[p] = hit ? is.prim : -1;
  }
  return 0;
}

void oracle_debug_bruteforce(int on) { g_debug_brute = on != 0; }

void oracle_fps(int op, const float* x, const float* y, int n, float* out) {
  for (int i = 0; i < n; ++i) {
    float a = x[i], b = y ? y[i] : 0.f, r = 0.f;
    switch (op) {
      case 0: r = fps::sin_(a); break;
      case 1: r = fps::cos_(a); break;
      case 2: r = fps::tan_(a); break;
      case 3: r = fps::exp_(a); break;
      case 4: r = fps::log_(a); break;
      case 5: r = fps::acos_(a); break;
      case 6: r = fps::asin_(a); break;
      case 7: r = fps::atan_(a); break;
      case 8: r = fps::pow_(a, b); break;
      case 9: r = fps::rsqrt_(a); break;
      case 10: r = fps::fmin_(a, b); break;
      case 11: r = fps::fmax_(a, b); break;
      case 12: r = (float)fps::f2i_(a); break;
    }
    out[i] = r;
  }
}

uint32_t oracle_wang_hash(uint32_t v) { return WangHash(v); }
void oracle_rng_floats(uint32_t seed, int n, float* out) {
  Rng r(seed);
  for (int i = 0; i < n; ++i) out[i] = r.uniform();
}
void oracle_rng_raw(uint32_t seed, int n, uint32_t* out) {
  Rng r(seed);
  for (int i = 0; i < n; ++i) out[i] = r.next();
}
void oracle_camera_ray(const oracle_ctx* c, uint32_t x, uint32_t y, uint32_t iter, float* o) {
  uint32_t pixel = x + y * (uint32_t)c->cx.scene.width;
  Rng rng(SampleSeed(pixel, iter));
  Ray r = CameraSample(c->cx, pixel, iter, rng);
  o[0] = r.origin.x; o[1] = r.origin.y; o[2] = r.origin.z;
  o[3] = r.dir.x; o[4] = r.dir.y; o[5] = r.dir.z;
}
int oracle_trace(const oracle_ctx* c, const float* o, const float* d, float tmin, float tmax, float* out) {
  Counters k;
  Ray r(v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]), -1, tmin, tmax);
  Isect is;
  if (!Intersect(c->cx, r, &is, k, RK_EXT)) return -1;
  out[0] = r.tmax;
  out[1] = is.pos.x; out[2] = is.pos.y; out[3] = is.pos.z;
  out[4] = is.n.x; out[5] = is.n.y; out[6] = is.n.z;
  out[7] = is.uv.x; out[8] = is.uv.y;
  out[9] = is.dpdu.x; out[10] = is.dpdu.y; out[11] = is.dpdu.z;
  return is.prim;
}
int oracle_occluded(const oracle_ctx* c, const float* o, const float* d, float tmin, float tmax) {
  Counters k;
  Ray r(v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]), -1, tmin, tmax);
  return IntersectPrimitive(c->cx, r, k, RK_SHADOW) ? 1 : 0;
}
int oracle_sample_camera(const oracle_ctx* c, const float* pos, float* out) {
  Ray ray;
  float we = 0.f, pdf = 0.f;
  int raster = -1;
  SampleCamera(c->cx, v3(pos[0], pos[1], pos[2]), ray, we, pdf, raster);
  out[0] = we; out[1] = pdf; out[2] = ray.tmax;
  return pdf != 0.f ? raster : -1;
}
void oracle_bsdf_sample(const oracle_ctx* c, int mat, const float* wo, const float* n, const float* dpdu,
                        const float* uv, const float* u3, float* o) {
  V3 out = v3(0, 0, 0), fr = v3(0, 0, 0);
  float pdf = 0.f;
  SampleBSDF(c->cx, c->cx.scene.materials[mat], v3(wo[0], wo[1], wo[2]), v3(n[0], n[1], n[2]), V2{uv[0], uv[1]},
             v3(dpdu[0], dpdu[1], dpdu[2]), v3(u3[0], u3[1], u3[2]), out, fr, pdf);
  o[0] = out.x; o[1] = out.y; o[2] = out.z; o[3] = fr.x; o[4] = fr.y; o[5] = fr.z; o[6] = pdf;
}
void oracle_bsdf_eval(const oracle_ctx* c, int mat, const float* wo, const float* wi, const float* n,
                      const float* dpdu, const float* uv, float* o) {
  V3 fr = v3(0, 0, 0);
  float pdf = 0.f;
  Fr(c->cx, c->cx.scene.materials[mat], v3(wo[0], wo[1], wo[2]), v3(wi[0], wi[1], wi[2]), v3(n[0], n[1], n[2]),
     V2{uv[0], uv[1]}, v3(dpdu[0], dpdu[1], dpdu[2]), fr, pdf);
  o[0] = fr.x; o[1] = fr.y; o[2] = fr.z; o[3] = pdf;
}
void oracle_light_sample(const oracle_ctx* c, int idx, const float* pos, const float* u2, float* o) {
  V3 rad, nor;
  Ray ray;
  float pdf;
  AreaSampleLight(c->cx, idx, v3(pos[0], pos[1], pos[2]), V2{u2[0], u2[1]}, rad, ray, nor, pdf);
  o[0] = rad.x; o[1] = rad.y; o[2] = rad.z;
  o[3] = ray.dir.x; o[4] = ray.dir.y; o[5] = ray.dir.z; o[6] = ray.tmax;
  o[7] = nor.x; o[8] = nor.y; o[9] = nor.z; o[10] = pdf;
}
int oracle_sample(oracle_ctx* c, uint32_t pixel, uint32_t iter, float* rgb) {
  Counters k;
  V3 col = v3(0, 0, 0);
  bool ok = Sample(c->cx, pixel, iter, col, k);
  rgb[0] = col.x; rgb[1] = col.y; rgb[2] = col.z;
  return ok ? 1 : 0;
}

}  // extern "C"
