// Floating-point contract shared by every implementation of the render path.
//
// The reference integrators call CUDA's sinf/cosf/expf/logf/acosf/atanf/tanf
// and rsqrtf (CUDA-PathTracer/tracer/Sampling.h:31-43, medium.h:13-38,
// sphere.h:74-80, pathtracer.cu:127-147, cutil_math.h:1197-1201).  Those are
// only specified to a few ulp, so two conforming builds of the reference
// already differ in the last bit.  This header pins every such function to
// one sequence of IEEE-754 binary32 add/mul/div/sqrt operations (translation
// units that include it are built with -ffp-contract=off; the only fused
// multiply-adds are the ones written out with fma_() below) so that an x86
// build and a gfx950 build produce the same bits for the same inputs.  On the domain the render path uses every
// function is within 3 ulp of the exact value (sin_/cos_ within 1e-13
// absolute next to their zeros, pow_ within 6 ulp); tests/test_fpspec.py
// checks that against double-precision libm.
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CASTER_HD __host__ __device__ inline
#else
#define CASTER_HD inline
#endif

namespace fps {

CASTER_HD uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
CASTER_HD float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// minNum / maxNum: a NaN operand is dropped, as CUDA's fminf/fmaxf do
// (reference slab test: CUDA-PathTracer/tracer/bbox.h:86-87).
CASTER_HD float fmin_(float a, float b) { return (a < b || b != b) ? a : b; }
CASTER_HD float fmax_(float a, float b) { return (a > b || b != b) ? a : b; }

// Stand-in for CUDA rsqrtf (cutil_math.h:1199): two correctly rounded ops.
CASTER_HD float rsqrt_(float x) { return 1.0f / sqrtf(x); }

CASTER_HD float clamp_(float v, float lo, float hi) { return fmin_(fmax_(v, lo), hi); }

// ---- fused multiply-add ----------------------------------------------------
// The reference is built with nvcc's default -fmad=true, so its own binary evaluates the dot
// and cross products of cutil_math.h:1132-1140, 1308-1311 with fused multiply-adds.  The
// translation units here are built with contraction OFF (an x86 compiler and the gfx950 compiler
// would not fuse the same operations), and the places that are fused are written out with
// fma_(), the same on both sides: exactly rounded on either machine (x86: vfmadd, the host
// objects are built with -mfma; gfx950: v_fma_f32), so the results stay equal bit for bit.
//   dot:    a.x*b.x + a.y*b.y + a.z*b.z   ->  fma(a.z, b.z, fma(a.x, b.x, a.y*b.y))
//   cross:  a*b - c*d                     ->  fma(a, b, -(c*d))
// which is how the compiler behind nvcc contracts those expressions (the first product of a sum
// becomes the multiplicand of the fma that adds the second; a difference of products keeps its
// second product rounded).
CASTER_HD float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
#ifdef CASTER_FPS_DIAG_UNFUSED_DOT   // diagnosis only: the products unfused
CASTER_HD float dot3_(float ax, float ay, float az, float bx, float by, float bz) { return ax * bx + ay * by + az * bz; }
CASTER_HD float diffprod_(float a, float b, float c, float d) { return a * b - c * d; }
#else
CASTER_HD float dot3_(float ax, float ay, float az, float bx, float by, float bz) {
  return fma_(az, bz, fma_(ax, bx, ay * by));
}
CASTER_HD float diffprod_(float a, float b, float c, float d) { return fma_(a, b, -(c * d)); }
#endif

// One plane of the slab test (bbox.h:76-84: (plane - origin) * (1 / direction)), for a ray that carries
// noi = -(origin * inv) per axis.  FUSED: fma(plane, inv, noi) -- one rounding of (plane - origin) * inv
// where the reference makes two; that form is monotone in `plane` like the reference's (a single rounding
// of a monotone function), so a box inside another is still entered no earlier and left no later than the
// outer one, which is all the traversal relies on.  A ray whose noi is not finite in some component (a
// direction component of zero, a non-finite origin, or origin * inv overflowing) keeps the reference's form
// for all its box tests: there inf - inf would make planes vanish from the test.
CASTER_HD bool slab_fused_(float noix, float noiy, float noiz) {
#ifdef CASTER_FPS_DIAG_UNFUSED_SLAB   // diagnosis only: every ray keeps the reference's form
  return false;
#endif
  return fabsf(noix) < INFINITY && fabsf(noiy) < INFINITY && fabsf(noiz) < INFINITY;
}
CASTER_HD float slab_plane_(float plane, float o, float inv, float noi, bool fused) {
  return fused ? fma_(plane, inv, noi) : (plane - o) * inv;
}

// Slack of the slab test's near-plane cull.  The reference drops a box whose entry distance
// exceeds the ray's tmax (CUDA-PathTracer/tracer/bbox.h:88-97: the far side is padded by
// 1 + 2*gamma(3), the near side is not).  Two triangles that share the edge a ray hits report
// the same distance, or distances a rounding apart, and once one of them has shortened the ray
// the other's box can fail that test by a rounding error of the slab arithmetic: which of the two
// is the closest hit then depends on the order the tree is walked in.  With the cull at
// tmax * kNearCullSlack both are always tested, and the result (smallest t, among equal t the
// highest primitive index) is the same for every traversal order -- which a wavefront traversal
// whose order depends on its neighbours needs in order to be a function of the ray alone.  It
// differs from the reference's own order-dependent choice in about one sample in 10^7 (a ray
// through a shared edge), by the difference between the two triangles' interpolated normals.
constexpr float kNearCullSlack = 1.0f + 1.0f / 65536.0f;

// float -> int with every out-of-range case defined (x86 and gfx950 disagree
// on NaN and overflow): NaN -> 0, saturating otherwise, truncation inside.
CASTER_HD int f2i_(float v) {
  if (v != v) return 0;
  if (v >= 2147483648.0f) return 2147483647;
  if (v <= -2147483648.0f) return -2147483647 - 1;
  return (int)v;
}

// 2^n as a float for n in [-126, 127].
CASTER_HD float pow2i_(int n) { return u2f((uint32_t)(n + 127) << 23); }

// ---- sin / cos -----------------------------------------------------------
// Octant reduction with a three-part pi/4 (products y*DP1, y*DP2 are exact
// for the |x| < 2^13 range used here), then degree-7/8 polynomials.
CASTER_HD void sincos_reduce_(float ax, int* jq, float* r) {
  int j = (int)(1.27323954473516f * ax);
  float y = (float)j;
  if (j & 1) { j += 1; y += 1.0f; }
  *jq = j & 7;
  *r = ((ax - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
}
CASTER_HD float sin_poly_(float x) {
  float z = x * x;
  return ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * x + x;
}
CASTER_HD float cos_poly_(float x) {
  float z = x * x;
  return ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z -
         0.5f * z + 1.0f;
}
CASTER_HD float sin_(float x) {
  bool neg = x < 0.0f;
  float ax = neg ? -x : x;
  int j; float r;
  sincos_reduce_(ax, &j, &r);
  if (j > 3) { neg = !neg; j -= 4; }
  float y = (j == 1 || j == 2) ? cos_poly_(r) : sin_poly_(r);
  return neg ? -y : y;
}
CASTER_HD float cos_(float x) {
  float ax = x < 0.0f ? -x : x;
  int j; float r;
  sincos_reduce_(ax, &j, &r);
  bool neg = false;
  if (j > 3) { j -= 4; neg = true; }
  if (j > 1) neg = !neg;
  float y = (j == 1 || j == 2) ? sin_poly_(r) : cos_poly_(r);
  return neg ? -y : y;
}
CASTER_HD float tan_(float x) {
  bool neg = x < 0.0f;
  float ax = neg ? -x : x;
  int j; float r;
  sincos_reduce_(ax, &j, &r);
  float z = r * r;
  float y = r;
  if (z > 1.0e-8f)
    y = (((((9.38540185543e-3f * z + 3.11992232697e-3f) * z + 2.44301354525e-2f) * z +
           5.34112807005e-2f) * z + 1.33387994085e-1f) * z + 3.33331568548e-1f) * z * r + r;
  if (j & 2) y = -1.0f / y;
  return neg ? -y : y;
}

// ---- exp -----------------------------------------------------------------
CASTER_HD float exp_(float x) {
  if (x != x) return x;
  if (x > 88.72283905206835f) return INFINITY;
  if (x < -103.278929903431851103f) return 0.0f;
  float z = floorf(1.44269504088896341f * x + 0.5f);
  x -= z * 0.693359375f;
  x -= z * -2.12194440e-4f;
  int n = (int)z;
  z = x * x;
  z = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x +
         4.1665795894e-2f) * x + 1.6666665459e-1f) * x + 5.0000001201e-1f) * z + x + 1.0f;
  // ldexp in two steps so that results in the subnormal range stay defined.
  int n1 = n / 2, n2 = n - n1;
  return z * pow2i_(n1) * pow2i_(n2);
}

// ---- log -----------------------------------------------------------------
CASTER_HD float log_(float x) {
  if (x != x || x < 0.0f) return NAN;
  if (x == 0.0f) return -INFINITY;
  if (x == INFINITY) return x;
  uint32_t u = f2u(x);
  int e = 0;
  if ((u >> 23) == 0) { x *= 8388608.0f; u = f2u(x); e = -23; }  // subnormal
  e += (int)(u >> 23) - 126;
  x = u2f((u & 0x007fffffu) | 0x3f000000u);  // mantissa in [0.5, 1)
  if (x < 0.707106781186547524f) { e -= 1; x = x + x - 1.0f; } else { x = x - 1.0f; }
  float z = x * x;
  float y = ((((((((7.0376836292e-2f * x - 1.1514610310e-1f) * x + 1.1676998740e-1f) * x -
                  1.2420140846e-1f) * x + 1.4249322787e-1f) * x - 1.6668057665e-1f) * x +
               2.0000714765e-1f) * x - 2.4999993993e-1f) * x + 3.3333331174e-1f) * x * z;
  float fe = (float)e;
  y += -2.12194440e-4f * fe;
  y += -0.5f * z;
  z = x + y;
  z += 0.693359375f * fe;
  return z;
}

CASTER_HD float pow_(float x, float y) { return exp_(y * log_(x)); }

// ---- asin / acos / atan ----------------------------------------------------
CASTER_HD float asin_(float x) {
  bool neg = x < 0.0f;
  float a = neg ? -x : x;
  if (a > 1.0f) return NAN;
  if (a < 1.0e-4f) return x;
  float z, t;
  bool flag = a > 0.5f;
  if (flag) { z = 0.5f * (1.0f - a); t = sqrtf(z); } else { t = a; z = t * t; }
  z = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z +
        7.4953002686e-2f) * z + 1.6666752422e-1f) * z * t + t;
  if (flag) { z = z + z; z = 1.5707963267948966192f - z; }
  return neg ? -z : z;
}
CASTER_HD float acos_(float x) {
  if (x < -1.0f || x > 1.0f) return NAN;
  if (x < -0.5f) return 3.14159265358979323846f - 2.0f * asin_(sqrtf(0.5f * (1.0f + x)));
  if (x > 0.5f) return 2.0f * asin_(sqrtf(0.5f * (1.0f - x)));
  return 1.5707963267948966192f - asin_(x);
}
CASTER_HD float atan_(float x) {
  bool neg = x < 0.0f;
  float a = neg ? -x : x;
  float y;
  if (a > 2.414213562373095f) { y = 1.5707963267948966192f; a = -(1.0f / a); }
  else if (a > 0.4142135623730950f) { y = 0.7853981633974483096f; a = (a - 1.0f) / (a + 1.0f); }
  else y = 0.0f;
  float z = a * a;
  y += (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z -
        3.33329491539e-1f) * z * a + a;
  return neg ? -y : y;
}

}  // namespace fps
