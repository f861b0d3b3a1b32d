// Part of the gfx950 wavefront renderer (included by ../kernels.hip, in this order: common.h, traverse.h, trace.h, shading.h, bsdf.h, path.h, volpath.h, lighttrace.h, bdpt.h, volpath_queues.h, film.h).
// Intersection records, materials and textures, lights and light selection (scattering: bsdf.h).
#pragma once

namespace caster {

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

// The diffuse texture is looked up once per vertex: its material is asked up to three times there
// (light sample, MIS sample, continuation) with the same uv, and each would repeat the four texel
// fetches.
__device__ inline void resolve_texture(const DevScene& S, Mat& m, V2 uv) {
  if (m.tex != -1 && (m.type == 0 || m.type == 5)) {  // the materials that read it: Lambertian, Substrate
    m.diffuse = albedo(S, m, uv);
    m.tex = -1;
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
// Area::SampleLight (area.h:12-17) with Triangle::SampleShape (Triangle.h:146-155): a uniform point on emissive
// triangle `idx` as seen from `pos`; one-sided, solid-angle density, shadow ray up to sqrt(d^2 - eps).
__device__ inline void sample_area_light(const DevScene& S, int idx, V3 pos, float ux, float uy, V3& radiance, V3& sdir,
                                         float& stmax, float& lightPdf, V3& normal) {
  LightRec l;
  load_light(S, idx, l);
  float su1 = sqrtf(ux);
  V2 uv = V2{1.f - su1, uy * su1};
  V3 p = uv.x * l.v0 + uv.y * l.v1 + (1 - uv.x - uv.y) * l.v2;
  normal = normalize(uv.x * l.n0 + uv.y * l.n1 + (1 - uv.x - uv.y) * l.n2);
  V3 dir = p - pos;
  float pdf = 1.f / (l.area * fabsf(dot(normal, normalize(dir)))) * dot(dir, dir);
  if (dot(normal, dir) >= 0.f) pdf = 0.f;
  lightPdf = pdf;
  radiance = pdf != 0.f ? l.radiance : v3(0.f, 0.f, 0.f);
  sdir = normalize(dir);
  stmax = sqrtf(dot(dir, dir) - S.epsilon);
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
    V3 normal;
    sample_area_light(S, idx, pos, ux, uy, radiance, sdir, stmax, lightPdf, normal);
  } else {
    float pdfW;
    V3 dir = uniform_sphere(ux, uy, pdfW);
    sdir = dir;
    stmax = 2.f * S.inf.radius - S.epsilon;
    lightPdf = pdfW;
    radiance = inf_le(S.inf, dir);
  }
}

}  // namespace caster
