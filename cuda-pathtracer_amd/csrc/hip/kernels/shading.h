// Part of the gfx950 wavefront renderer (included by ../kernels.hip, in this order: common.h, traverse.h, trace.h, shading.h, path.h, volpath.h, lighttrace.h, bdpt.h, volpath_queues.h, film.h).
// Intersection records, materials and textures, BSDF sample/eval, lights and light selection.
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

}  // namespace caster
