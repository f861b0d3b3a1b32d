// Part of the gfx950 wavefront renderer (included by ../kernels.hip, in this order: common.h, traverse.h, trace.h, shading.h, bsdf.h, path.h, volpath.h, lighttrace.h, bdpt.h, volpath_queues.h, film.h).
// Bidirectional path tracing (k_bdpt).
#pragma once

namespace caster {

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
// The shadow ray of a join, when it is not walked where the join is evaluated.
struct JoinRay {
  bool want;
  V3 o, d;
  float tmax;
  int medium;
};
#ifdef CASTER_BDPT_DIAG_NO_TR   // diagnosis builds only: every join sees its other end
#define transmittance(...) v3(1.f, 1.f, 1.f)
#endif
// QUEUED: the transmittance between the joined vertices is not walked here; the join comes back with transmittance
// one and its shadow ray in `ray` (which is wanted wherever the reference calls Tr(), whatever the join turns out to
// be worth afterwards: the ray statistics are the reference's), and the caller sends the ray through the traversal
// kernel and multiplies afterwards.
template <bool QUEUED>
__device__ V3 bdpt_connect(const DevScene& S, Stack& st, Stats* stats, const BdptStore& V, uint32_t g, int s, int t, int& raster,
                           uint32_t& rng, uint32_t& n_tr, JoinRay& ray) {
  const V3 zero = v3(0.f, 0.f, 0.f);
  const int LB = kBdptPathVerts;
  const float eps = S.epsilon;
  auto visibility = [&](V3 o, V3 d, float tmax, int medium) -> V3 {
    if (QUEUED) {
      ray.want = true; ray.o = o; ray.d = d; ray.tmax = tmax; ray.medium = medium;
      return v3(1.f, 1.f, 1.f);
    }
    return transmittance(S, o, d, eps, tmax, medium, st, stats, n_tr, rng);
  };
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
    V3 tr = visibility(cur.pos, sdir, stmax, cur.medium);
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
    V3 tr = visibility(cur.pos, sdir, stmax, cur.medium);
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
  V3 tr = visibility(c1.pos, c1Tol1, stmax, c1.medium);
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

#ifdef CASTER_BDPT_DIAG_NO_TR
#undef transmittance
#endif

// ---- joins whose shadow rays go through the traversal kernel ---------------------------------------------------------
// On a scene of a million triangles nine tenths of the one-kernel integrator's time is tree walks, every lane in a loop
// of its own, and six tenths of it the joins' shadow rays.  Where Tr() draws no random numbers (no density grid) a
// join's contribution is its value without the transmittance times the transmittance, so the wavefront that evaluates
// a join only writes down the value and the ray; the rays of a launch are advanced a Tr() segment per pass of the
// traversal kernel (k_tr_step / k_tr_finish, as the volumetric integrator's shadow rays are) and k_bdpt_resolve
// splats value x transmittance.  Join slot i is entry i of the queue.
//
// Slots are handed out a round at a time: when its 64 samples have been walked a wavefront knows how many joins they
// have that can want a shadow ray (all but the emitter hits) and takes that many slots with one compare-and-swap; join
// j of the round then owns slot first + j, and a join that turns out to want no ray fills its slot with a ray that
// is over before it starts.  A round that finds the queue too full is put on a list instead and evaluated afterwards by
// the kernel that walks its shadow rays in place -- so no launch depends on a bound on the joins of a sample, and the
// kernel that queues carries no tree walk for its joins (with one it ran a third slower: 139 against 181 Msamples/s).
struct BdptJoins {
  RayQueue q;            // the rays, in slot order (q.slot[i] == i)
  float4* val;           // slot -> value without transmittance | raster index
  uint32_t cap;          // slots
  uint32_t* unused;      // slots of this launch filled with a ray that is over before it starts
  uint32_t* deferred;    // first samples of the rounds left to the in-place kernel
  uint32_t* n_deferred;
};
// After a launch of k_bdpt<true>: the slots it handed out; the empty ones come off the count of transmittance segments
// that the first traversal pass is about to add.
__global__ void k_join_count(uint32_t* count, uint32_t cap, uint32_t* n_out, uint32_t* unused, Stats* stats) {
  const uint32_t n = *count < cap ? *count : cap;   // (rounds that found no room left the counter beyond the queue)
  *count = n;
  *n_out = n;
  stats->rays[3] -= (unsigned long long)*unused;
  *unused = 0u;
}
__global__ __launch_bounds__(256) void k_bdpt_resolve(LaunchCtx L, BdptJoins J, const uint32_t* n_ptr, float* acc) {
  const uint32_t n = *n_ptr;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float4 v = J.val[i];
    const V3 Lc = v3(v) * v3(L.P.pend_w[i]);
    if (has_inf(Lc) || has_nan(Lc) || is_black(Lc)) continue;
    splat(L.S, acc, __float_as_int(v.w), Lc);
  }
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
#ifndef CASTER_BDPT_WAVES
// Wavefronts per SIMD the register budget is held to.  Left to itself the kernel takes 189 registers (two wavefronts
// per SIMD) and waits for memory most of the time; held to 96 it spills 287 values to scratch and is still half as
// fast again, because five wavefronts cover one another's waits (benchmark scene, 1080p: 65 Msamples/s at 189
// registers, 83 at 168, 99 at 128, 105 at 96, 92 at 80).
#define CASTER_BDPT_WAVES 5
#endif
// (the kernel that queues its joins' shadow rays carries no tree walk in its join loop and does best with the budget of
// four: 128 registers; benchmark scene 191 -> 194 Msamples/s, staircase 53 -> 56, Cornell box 102 -> 117)
#ifndef CASTER_BDPT_QUEUED_WAVES
#define CASTER_BDPT_QUEUED_WAVES 4
#endif
// Samples [begin, end) of the batch (multiples of the wavefront size), or -- `rounds` given -- the rounds of 64 samples
// that start at the *n_rounds sample indices listed there.  QUEUED: see BdptJoins.
template <bool QUEUED>
__global__ __launch_bounds__(kWave, QUEUED ? CASTER_BDPT_QUEUED_WAVES : CASTER_BDPT_WAVES) void k_bdpt(LaunchCtx L, uint32_t first_iter, uint32_t n_iter, uint32_t begin,
                                                                    uint32_t end, float* acc, WalkAux aux, BdptStore V, BdptJoins J,
                                                                    const uint32_t* rounds, const uint32_t* n_rounds) {
  __shared__ uint2 s_stack[kLdsDepth][kWave];
  __shared__ int s_pix[kWave];
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
  uint32_t n_dummy = 0;
  const uint32_t n_round = rounds ? *n_rounds : (end - begin) / kWave;
  for (uint32_t round = blockIdx.x; round < n_round; round += gridDim.x) {   // (the loop is wave-uniform)
    const uint32_t idx0 = rounds ? rounds[round] : begin + round * kWave, idx = idx0 + lane;
    const uint32_t n_ext_before = n_ext;
    uint32_t k, local;
    batch_slot_split(S, idx, n_iter, k, local);
    uint32_t px, py;
    const bool live = slot_pixel(S, local, px, py);
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
          JoinRay unused;
          V3 Lc = bdpt_connect<false>(S, st, L.stats, V, g, s, t, raster, rng, n_tr, unused);
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
    s_pix[lane] = (int)pixel;
    uint32_t slot_next = 0;   // QUEUED: the slot of the first join of the kind under way
    if (QUEUED) {
      const int cs = nCamera - 1 > 0 ? nCamera - 1 : 0, lt = nLight - 1 > 0 ? nLight - 1 : 0;
      uint32_t want = (uint32_t)((nLight >= 1 ? cs : 0) + (nCamera >= 1 ? lt : 0) + cs * lt);
      for (int off = 32; off > 0; off >>= 1) want += (uint32_t)__shfl_xor((int)want, off, kWave);
      // (one atomic add per round, never a compare-and-swap loop: some thousand wavefronts retrying on one address
      // made the launch six times longer)
      // (a queue that is full already is not added to: the counter then stays within one round per wavefront of the
      // queue's length and cannot wrap around, however many joins the samples of a launch have)
      uint32_t first = 0u;
      if (want != 0 && lane == 0) {
        first = __atomic_load_n(J.q.count, __ATOMIC_RELAXED);
        if (first < J.cap) first = atomicAdd(J.q.count, want);
      }
      first = (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
      if (want != 0 && (first >= J.cap || want > J.cap - first)) {
        // no room: the round is left to the kernel that walks its shadow rays in place; what it was given of the queue's
        // end is filled with rays that are over before they start
        const uint32_t room = first < J.cap ? J.cap - first : 0u;
        for (uint32_t i = lane; i < room; i += kWave) {
          const uint32_t slot = first + i;
          J.q.o[slot] = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
          J.q.d[slot] = make_float4(0.f, 0.f, 1.f, -1.f);
          J.q.slot[slot] = (int)slot;
          L.P.pend_w[slot] = f4(v3(1.f, 1.f, 1.f), (uint32_t)-1);
          J.val[slot] = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
        }
        n_dummy += room;
        if (lane == 0) J.deferred[atomicAdd(J.n_deferred, 1u)] = idx0;
        n_ext = n_ext_before;
        continue;
      }
      slot_next = first;
    }
#ifdef CASTER_BDPT_DIAG_KINDS   // diagnosis builds only: a bit per kind of join that is evaluated
    for (int kind = 0; kind < 4; ++kind) if ((CASTER_BDPT_DIAG_KINDS >> kind) & 1) {
#else
    for (int kind = 0; kind < 4; ++kind) {
#endif
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
      for (uint32_t j0 = 0; j0 < wave_joins; j0 += kWave) {
        const uint32_t j = j0 + lane;
        JoinRay ray;
        ray.want = false; ray.o = v3(0, 0, 0); ray.d = v3(0, 0, 1); ray.tmax = 0.f; ray.medium = -1;
        V3 val = v3(0.f, 0.f, 0.f);
        int target = -1;
        if (j < wave_joins) {
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
          V3 Lc = bdpt_connect<QUEUED>(S, st, L.stats, V, g0 + i, s, t, raster, jr, n_tr, ray);
          if (QUEUED && ray.want) {
            val = Lc;
            target = s == 1 ? raster : s_pix[i];
          } else if (!(has_inf(Lc) || has_nan(Lc) || is_black(Lc))) {
            if (s == 1) {
              splat(S, acc, raster, Lc);
            } else {  // gathered in LDS: one add to the film per sample instead of one per join
              atomicAdd(&s_own[i][0], Lc.x); atomicAdd(&s_own[i][1], Lc.y); atomicAdd(&s_own[i][2], Lc.z);
            }
          }
        }
        if (QUEUED && kind != 0) {
          if (j < wave_joins) {
            const uint32_t slot = slot_next + j;
            // (a join that wants no ray: one that is over before it starts, tmax < 0)
            J.q.o[slot] = f4(ray.o, __int_as_float(-1));
            J.q.d[slot] = f4(ray.d, ray.want ? ray.tmax : -1.f);
            J.q.slot[slot] = (int)slot;
            L.P.pend_w[slot] = f4(v3(1.f, 1.f, 1.f), (uint32_t)ray.medium);
            J.val[slot] = f4(val, __int_as_float(target));
          }
          n_dummy += (uint32_t)__popcll(lane_ballot(j < wave_joins && !ray.want));
        }
      }
      if (QUEUED && kind != 0) slot_next += wave_joins;
    }
    __syncthreads();
    {
      V3 own = v3(s_own[lane][0], s_own[lane][1], s_own[lane][2]);
      if (live && !is_black(own)) splat(S, acc, (int)pixel, own);
    }
  }
  if (QUEUED && lane == 0 && n_dummy) atomicAdd(J.unused, n_dummy);
  for (int off = 32; off > 0; off >>= 1) {
    n_tr += (uint32_t)__shfl_down((int)n_tr, off, 64);
    n_ext += (uint32_t)__shfl_down((int)n_ext, off, 64);
  }
  if ((threadIdx.x & 63u) == 0) {
    if (n_tr) atomicAdd(&L.stats->rays[3], (unsigned long long)n_tr);
    if (n_ext) atomicAdd(&L.stats->rays[0], (unsigned long long)n_ext);
  }
}

}  // namespace caster
