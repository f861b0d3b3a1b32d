// Part of the gfx950 wavefront renderer (included by ../kernels.hip after shading.h).
// Scattering at a surface vertex: one "lobe" object per material type.
//
// The reference evaluates a vertex's material three times -- Fr() for the light sample, SampleBSDF()
// for the light probe, SampleBSDF() again for the continuation (pathtracer.cu:937, :945, :991) --
// each a trip through a six-way switch that starts from scratch (pathtracer.cu:500-818).  Here a
// vertex builds ONE lobe of its material's type from (direction to the previous vertex, normal,
// tangent, material) and asks it up to three times.  What the three answers share is computed once,
// in init(): the normal on the viewer's side and the tangent frame, the albedo (a texture lookup),
// the viewer's Smith shadowing term, the Substrate's view-dependent diffuse factor, the ordered
// indices of refraction.  eval() and sample() of the microfacet lobes end in the same `finish`
// function, so the wavefront runs one copy of that code whichever way the half vector was found.
// A shade wavefront handles vertices of ONE material type at a time (k_shade_pt sorts them), so
// the lobe type is a compile-time choice there; the other integrators go through the dispatching
// wrappers at the end of this file.
//
// Every float operation and its order are those of the reference's formulas (quirks included:
// the (1 - cos phi) term of GGX_D, alphaU used twice in the anisotropic SampleGGX), because the
// film must equal the CPU oracle's bit for bit; hoisting only moves an operation, never changes it.
#pragma once

namespace caster {

// Material type numbers of the scene description (material.h:6-13).
enum : int { kMatLambert = 0, kMatMirror = 1, kMatDielectric = 2, kMatRoughDielectric = 3, kMatRoughConductor = 4, kMatSubstrate = 5 };

__device__ inline bool is_delta(int t) { return t == kMatMirror || t == kMatDielectric; }  // material.h:33-35
__device__ inline float power_heuristic(float fPdf, float gPdf) {   // pathtracer.cu:175-178 with nf = ng = 1
  float f = 1 * fPdf, g = 1 * gPdf;
  return (f * f) / (f * f + g * g);
}
__device__ inline float luminance(V3 c) { return dot(c, v3(0.212671f, 0.715160f, 0.072169f)); }
__device__ inline V3 mirror_about(V3 w, V3 axis) { return 2.f * dot(w, axis) * axis - w; }   // pathtracer.cu:149
__device__ inline float pow5(float v) { return (v * v) * (v * v) * v; }
__device__ inline V3 polar_dir(float s, float c, float phi) { return v3(s * fps::cos_(phi), s * fps::sin_(phi), c); }

// Unpolarised Fresnel reflectance of a dielectric boundary (pathtracer.cu:50-56).
__device__ inline float fresnel_dielectric(float cosi, float cost, float etai, float etat) {
  float par = (etat * cosi - etai * cost) / (etat * cosi + etai * cost);
  float per = (etai * cosi - etat * cost) / (etai * cosi + etat * cost);
  return (par * par + per * per) * 0.5f;
}
// ... of a conductor (pathtracer.cu:58-66).
__device__ inline V3 fresnel_conductor(float cosi, V3 eta, V3 k) {
  V3 ek = eta * eta + k * k;
  V3 a = ek * cosi * cosi;
  V3 par = (a - eta * cosi * 2.f + 1.f) / (a + eta * cosi * 2.f + 1.f);
  V3 per = (ek - eta * cosi * 2.f + cosi * cosi) / (ek + eta * cosi * 2.f + cosi * cosi);
  return (par + per) * 0.5f;
}

// What a vertex knows before any direction is chosen.
struct Surface {
  V3 wo;  // unit direction towards the previous vertex (the reference's `in`)
  V3 n;   // normal of the intersection record, as is
  V3 tu;  // its tangent (dpdu)
};

// The tangent frame (tu, tu x m, m) around a normal m; not re-orthogonalised, as in the reference.
struct Frame {
  V3 tu, tv, m;
  __device__ void set(V3 tangent, V3 normal) { tu = tangent; m = normal; tv = cross(tangent, normal); }
  __device__ V3 world(V3 local) const { return to_world(local, tu, tv, m); }
};

// GGX microfacet distribution around `m` with the reference's D, Smith term and sampling
// (pathtracer.cu:68-147).
struct Ggx {
  float au, av;
  V3 m, tu;
  __device__ float d(V3 wh) const {
    float c = dot(wh, m);
    if (c <= 0.f) return 0.f;
    c = fps::clamp_(c, 0.f, 1.f);
    float c2 = c * c;
    float s2 = 1.f - c2;
    float c4 = c2 * c2;
    float t2 = s2 / c2;
    V3 flat = normalize(wh - c * m);
    float cphi = dot(flat, tu);
    float e = t2 * (cphi * cphi / (au * au) + (1.f - cphi) / (av * av));
    return 1.f / (PI * au * av * c4 * (1 + e) * (1 + e));
  }
  // Smith's G1 for direction w is zero when w and the half vector disagree about the side, and
  // otherwise this value, which does not depend on the half vector: a vertex computes it once for
  // its viewer direction.  `wm` = dot(w, m).
  __device__ float g1_value(V3 w, float wm) const {
    float s = sqrtf(fps::clamp_(1.f - wm * wm, 0.f, 1.f));
    float t = s / wm;
    if (isinf_(t)) return 0.f;
    V3 flat = normalize(w - wm * m);
    float cphi = dot(flat, tu);
    float c2 = cphi * cphi;
    float s2 = 1.f - c2;
    float a2 = c2 * (au * au) + s2 * (av * av);
    float a2t2 = a2 * t * t;
    return 2.f / (1.f + sqrtf(1 + a2t2));
  }
  __device__ float g1(V3 w, V3 wh) const {
    float wm = dot(w, m);
    if (wm * dot(w, wh) < 0.f) return 0.f;
    return g1_value(w, wm);
  }
  // A half vector in the local frame.
  __device__ V3 sample_local(float u1, float u2) const {
    if (au == av) {
      float c = sqrtf((1.f - u1) / (u1 * (au * av - 1.f) + 1.f));
      float s = sqrtf(1.f - c * c);
      return polar_dir(s, c, TWOPI * u2);
    }
    float phi = fps::atan_(av / au * fps::tan_(TWOPI * u2 + .5f * PI));
    if (u2 > .5f) phi += PI;
    float sphi = fps::sin_(phi), cphi = fps::cos_(phi);
    const float au2 = au * au;
    const float a2 = 1.0f / (cphi * cphi / au2 + sphi * sphi / au2);
    float t2 = a2 * u1 / (1.f - u1);
    float c = 1 / sqrtf(1 + t2);
    float s = sqrtf(fps::fmax_(0.f, 1.f - c * c));
    return polar_dir(s, c, phi);
  }
};

// The viewer's side of Smith's term, kept per vertex.
struct ViewerG1 {
  float wm, value;
  __device__ void set(const Ggx& g, V3 wo) { wm = dot(wo, g.m); value = g.g1_value(wo, wm); }
  __device__ float at(V3 wo, V3 wh) const { return wm * dot(wo, wh) < 0.f ? 0.f : value; }
};

// ------------------------------------------------------------------ Lambertian (pathtracer.cu:502-513, 702-711)
struct LambertLobe {
  static constexpr bool kDelta = false;
  V3 wo, n, fr_const;
  Frame f;
  __device__ void init(const Surface& s, const Mat& m, V3 albedo_rgb) {
    wo = s.wo; n = s.n;
    f.set(s.tu, dot(s.n, s.wo) < 0 ? -s.n : s.n);
    fr_const = albedo_rgb * ONE_OVER_PI;
  }
  __device__ void eval(V3 wi, V3& fr, float& pdf) const {
    if (!(dot(wo, n) * dot(wi, n) > 0)) { fr = v3(0, 0, 0); pdf = 0.f; return; }
    fr = fr_const;
    pdf = fabsf(dot(wi, n)) * ONE_OVER_PI;
  }
  __device__ void sample(V3 u, bool, V3& wi, V3& fr, float& pdf) const {
    wi = f.world(cosine_hemisphere(u.x, u.y, pdf));
    fr = fr_const;
  }
};

// ------------------------------------------------------------------ Mirror (pathtracer.cu:515-519)
struct MirrorLobe {
  static constexpr bool kDelta = true;
  V3 wo, n, ks;
  __device__ void init(const Surface& s, const Mat& m, V3) { wo = s.wo; n = s.n; ks = m.specular; }
  __device__ void eval(V3, V3& fr, float& pdf) const { fr = v3(0, 0, 0); pdf = 0.f; }
  __device__ void sample(V3, bool, V3& wi, V3& fr, float& pdf) const {
    wi = mirror_about(wo, n);
    fr = ks / fabsf(dot(wi, n));
    pdf = 1.f;
  }
};

// ------------------------------------------------------------------ smooth dielectric (pathtracer.cu:521-560, 153-167)
struct DielectricLobe {
  static constexpr bool kDelta = true;
  V3 wo, n, ks;
  float eta_out, eta_in;
  __device__ void init(const Surface& s, const Mat& m, V3) {
    wo = s.wo; n = s.n; ks = m.specular; eta_out = m.outsideIOR; eta_in = m.insideIOR;
  }
  __device__ void eval(V3, V3& fr, float& pdf) const { fr = v3(0, 0, 0); pdf = 0.f; }
  // The refracted direction (Refract(): its own ordering of the two indices and its own cosine).
  __device__ V3 bent() const {
    float c = dot(wo, n);
    bool enter = c > 0;
    float a = enter ? eta_out : eta_in, b = enter ? eta_in : eta_out;
    float eta = a / b;
    float st2 = (1.f - c * c) * eta * eta;
    float ct = sqrtf(1.f - st2);
    return normalize((n * c - wo) * eta + (enter ? -ct : ct) * n);
  }
  __device__ void sample(V3 u, bool radiance, V3& wi, V3& fr, float& pdf) const {
    float cosi = dot(-wo, n);
    float ei = eta_out, et = eta_in;
    if (cosi > 0) { float t = et; et = ei; ei = t; }
    float eta = ei / et;
    float sint2 = eta * eta * (1.f - cosi * cosi);
    if (sint2 > 1.f) {   // total internal reflection
      wi = mirror_about(wo, n);
      fr = ks / fabsf(dot(wi, n));
      pdf = 1.f;
      return;
    }
    float cost = sqrtf(1.f - sint2 < 0.f ? 0.f : 1.f - sint2);
    float F = fresnel_dielectric(fabsf(cost), fabsf(cosi), et, ei);
    if (u.x > F) {
      wi = bent();
      fr = ks / fabsf(dot(wi, n)) * (1.f - F);
      if (radiance) fr = fr * (eta * eta);
      pdf = 1.f - F;
    } else {
      wi = mirror_about(wo, n);
      fr = ks / fabsf(dot(wi, n)) * F;
      pdf = F;
    }
  }
};

// ------------------------------------------------------------------ rough conductor (pathtracer.cu:562-588, 723-742)
struct RoughConductorLobe {
  static constexpr bool kDelta = false;
  V3 wo, n, ks, eta, k;
  Frame f;
  Ggx g;
  ViewerG1 gv;
  float abs_wo_m;
  __device__ void init(const Surface& s, const Mat& m, V3) {
    wo = s.wo; n = s.n; ks = m.specular; eta = m.eta; k = m.k;
    f.set(s.tu, dot(s.n, s.wo) < 0 ? -s.n : s.n);
    g.au = m.alphaU; g.av = m.alphaV; g.m = f.m; g.tu = s.tu;
    gv.set(g, wo);
    abs_wo_m = fabsf(gv.wm);
  }
  // Reflection off half vector wh into wi.
  __device__ void finish(V3 wh, V3 wi, V3& fr, float& pdf) const {
    if (!(dot(wo, n) * dot(wi, n) > 0)) { fr = v3(0, 0, 0); pdf = 0.f; return; }
    V3 F = fresnel_conductor(fabsf(dot(wi, wh)), eta, k);
    float D = g.d(wh);
    float G = gv.at(wo, wh) * g.g1(wi, wh);
    fr = ks * F * D * G / (4.f * abs_wo_m * fabsf(dot(wi, g.m)));
    pdf = D * fabsf(dot(wh, g.m)) / (4.f * fabsf(dot(wo, wh)));
  }
  __device__ void eval(V3 wi, V3& fr, float& pdf) const { finish(normalize(wo + wi), wi, fr, pdf); }
  __device__ void sample(V3 u, bool, V3& wi, V3& fr, float& pdf) const {
    V3 wh = f.world(g.sample_local(u.x, u.y));
    wi = mirror_about(wo, wh);
    finish(wh, wi, fr, pdf);
  }
};

// ------------------------------------------------------------------ substrate (pathtracer.cu:590-640, 744-775)
struct SubstrateLobe {
  static constexpr bool kDelta = false;
  V3 wo, ks, diffuse_const;
  Frame f;
  Ggx g;
  float cos_o, fade_o;
  __device__ void init(const Surface& s, const Mat& m, V3 albedo_rgb) {
    wo = s.wo; ks = m.specular;
    f.set(s.tu, dot(s.n, s.wo) < 0 ? -s.n : s.n);
    g.au = m.alphaU; g.av = m.alphaV; g.m = f.m; g.tu = s.tu;
    cos_o = fabsf(dot(wo, f.m));
    fade_o = 1 - pow5(1 - .5f * cos_o);
    diffuse_const = (28.f / (23.f * PI)) * albedo_rgb * (v3(1.f, 1.f, 1.f) - ks);
  }
  __device__ void finish(V3 wi, V3& fr, float& pdf) const {
    if (!(dot(wo, f.m) * dot(wi, f.m) > 0)) { fr = v3(0, 0, 0); pdf = 0.f; return; }
    float cos_i = fabsf(dot(wi, f.m));
    V3 diffuse = diffuse_const * (1 - pow5(1 - .5f * cos_i)) * fade_o;
    V3 wh = normalize(wo + wi);
    float D = g.d(wh);
    float larger = cos_i > cos_o ? cos_i : cos_o;
    float wi_h = dot(wi, wh);
    V3 glossy = D / (4.f * fabsf(wi_h) * larger) * (ks + pow5(1.f - wi_h) * (1 - ks));
    fr = diffuse + glossy;
    pdf = 0.5f * (cos_i * ONE_OVER_PI + D * fabsf(dot(wh, f.m)) / (4.f * dot(wo, wh)));
  }
  __device__ void eval(V3 wi, V3& fr, float& pdf) const { finish(wi, fr, pdf); }
  __device__ void sample(V3 u, bool, V3& wi, V3& fr, float& pdf) const {
    if (u.x < 0.5f) {
      float unused;
      wi = f.world(cosine_hemisphere(fps::fmin_(2.f * u.x, 0x1.fffffep-1f), u.y, unused));
    } else {
      V3 wh = f.world(g.sample_local(fps::fmin_(2 * (u.x - 0.5f), 0x1.fffffep-1f), u.y));
      wi = mirror_about(wo, wh);
    }
    finish(wi, fr, pdf);
  }
};

// ------------------------------------------------------------------ rough dielectric (pathtracer.cu:642-696, 777-816)
struct RoughDielectricLobe {
  static constexpr bool kDelta = false;
  V3 wo, ks;
  Frame f;   // around the normal as is: this material does not turn it towards the viewer
  Ggx g;
  ViewerG1 gv;
  float ei, et, eta, cos_view;   // indices ordered by the side the viewer is on; cos_view = dot(-wo, n)
  __device__ void init(const Surface& s, const Mat& m, V3) {
    wo = s.wo; ks = m.specular;
    f.set(s.tu, s.n);
    g.au = m.alphaU; g.av = m.alphaV; g.m = s.n; g.tu = s.tu;
    gv.set(g, wo);
    ei = m.outsideIOR; et = m.insideIOR;
    cos_view = dot(-wo, s.n);
    if (cos_view > 0) { float t = et; et = ei; ei = t; }
    eta = ei / et;
  }
  // Fresnel term at half vector wh; `tir` when nothing is transmitted.
  __device__ float fresnel_at(V3 wh, float& cos_h, float& cos_t, bool& tir) const {
    cos_h = dot(-wo, wh);
    float s2i = fps::fmax_(0.f, 1.f - cos_h * cos_h);
    float s2t = eta * eta * s2i;
    cos_t = sqrtf(1.f - s2t < 0.f ? 0.f : 1.f - s2t);
    tir = s2t > 1.f;
    return fresnel_dielectric(fabsf(cos_h), fabsf(cos_t), ei, et);
  }
  __device__ void reflected(V3 wh, V3 wi, float D, float weight, bool weighted, V3& fr, float& pdf) const {
    float G = gv.at(wo, wh) * g.g1(wi, wh);
    float denom = 4.f * fabsf(gv.wm) * fabsf(dot(wi, g.m));
    float p = D * fabsf(dot(wh, g.m)) / (4.f * fabsf(dot(wh, wo)));
    if (weighted) { fr = ks * weight * D * G / denom; }
    else { fr = ks * D * G / denom; }
    pdf = p;
  }
  __device__ void transmitted(V3 wh, V3 wi, float D, float F, bool radiance, V3& fr, float& pdf) const {
    float G = gv.at(wo, wh) * g.g1(wi, wh);
    float wi_h = dot(wi, wh), wo_h = dot(wo, wh);
    float c = et * wi_h + ei * wo_h;
    fr = ks * ei * ei * D * G * (1.f - F) * fabsf(wo_h) * fabsf(wi_h) / (fabsf(dot(wi, g.m)) * fabsf(gv.wm) * c * c);
    if (radiance) fr = fr * (1.f / (eta * eta));
    pdf = (1.f - F) * D * fabsf(dot(wh, g.m)) * et * et * fabsf(wi_h) / (c * c);
  }
  __device__ void eval(V3 wi, V3& fr, float& pdf) const {
    bool same_side = dot(wo, g.m) * dot(wi, g.m) > 0;
    V3 wh = normalize(-(ei * wo + et * wi));
    float D = g.d(wh);
    float cos_h, cos_t;
    bool tir;
    float F = fresnel_at(wh, cos_h, cos_t, tir);
    if (!same_side) {
      transmitted(wh, wi, D, F, true, fr, pdf);
    } else {
      reflected(wh, wi, D, F, true, fr, pdf);
      pdf = F * D * fabsf(dot(wh, g.m)) / (4.f * fabsf(dot(wh, wo)));
    }
  }
  __device__ void sample(V3 u, bool radiance, V3& wi, V3& fr, float& pdf) const {
    V3 wh = f.world(g.sample_local(u.x, u.y));
    float D = g.d(wh);
    float cos_h, cos_t;
    bool tir;
    float F = fresnel_at(wh, cos_h, cos_t, tir);
    if (tir) {
      wi = mirror_about(wo, wh);
      reflected(wh, wi, D, 1.f, false, fr, pdf);
      return;
    }
    if (u.z > F) {
      wi = normalize((-wo - wh * cos_h) * eta + (cos_view < 0 ? -cos_t : cos_t) * wh);
      transmitted(wh, wi, D, F, radiance, fr, pdf);
    } else {
      wi = mirror_about(wo, wh);
      reflected(wh, wi, D, F, true, fr, pdf);
      pdf = pdf * F;
    }
  }
};

// ------------------------------------------------------------------ by material type number
template <int T> struct LobeOf;
template <> struct LobeOf<kMatLambert> { typedef LambertLobe type; };
template <> struct LobeOf<kMatMirror> { typedef MirrorLobe type; };
template <> struct LobeOf<kMatDielectric> { typedef DielectricLobe type; };
template <> struct LobeOf<kMatRoughDielectric> { typedef RoughDielectricLobe type; };
template <> struct LobeOf<kMatRoughConductor> { typedef RoughConductorLobe type; };
template <> struct LobeOf<kMatSubstrate> { typedef SubstrateLobe type; };

// Calls fn(lobe) with the lobe of material `m` built at surface `s`; the materials that read the
// diffuse colour get it resolved (texture or constant) first.
template <class Fn>
__device__ inline void with_lobe(const DevScene& S, const Mat& m, const Surface& s, V2 uv, Fn&& fn) {
  switch (m.type) {
    case kMatLambert: { LambertLobe l; l.init(s, m, albedo(S, m, uv)); fn(l); break; }
    case kMatMirror: { MirrorLobe l; l.init(s, m, v3(0, 0, 0)); fn(l); break; }
    case kMatDielectric: { DielectricLobe l; l.init(s, m, v3(0, 0, 0)); fn(l); break; }
    case kMatRoughDielectric: { RoughDielectricLobe l; l.init(s, m, v3(0, 0, 0)); fn(l); break; }
    case kMatRoughConductor: { RoughConductorLobe l; l.init(s, m, v3(0, 0, 0)); fn(l); break; }
    case kMatSubstrate: { SubstrateLobe l; l.init(s, m, albedo(S, m, uv)); fn(l); break; }
    default: break;
  }
}

// One-off sample / evaluation for the integrators that look at a vertex once at a time (light
// tracing, the bidirectional joins, the volumetric integrator): SampleBSDF and Fr of the reference
// (pathtracer.cu:500-698, 700-818).  `radiance` false is TransportMode::Importance (light paths),
// which drops the radiance scaling of refraction (:549-550, :683-684).
__device__ inline void sample_bsdf(const DevScene& S, const Mat& m, V3 in, V3 nor, V2 uv, V3 dpdu, V3 u, V3& out, V3& fr,
                                   float& pdf, bool radiance = true) {
  out = v3(0, 0, 1); fr = v3(0, 0, 0); pdf = 0.f;
  with_lobe(S, m, Surface{in, nor, dpdu}, uv, [&](const auto& lobe) { lobe.sample(u, radiance, out, fr, pdf); });
}
__device__ inline void eval_bsdf(const DevScene& S, const Mat& m, V3 in, V3 out, V3 nor, V2 uv, V3 dpdu, V3& fr, float& pdf) {
  fr = v3(0, 0, 0); pdf = 0.f;
  with_lobe(S, m, Surface{in, nor, dpdu}, uv, [&](const auto& lobe) { lobe.eval(out, fr, pdf); });
}

}  // namespace caster
