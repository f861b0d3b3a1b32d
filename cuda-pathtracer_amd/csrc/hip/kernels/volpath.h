// Part of the gfx950 wavefront renderer (included by ../kernels.hip, in this order: common.h, traverse.h, trace.h, shading.h, bsdf.h, path.h, volpath.h, lighttrace.h, bdpt.h, volpath_queues.h, film.h).
// Media, Tr(), the volumetric vertex with in-kernel walks (k_shade_vpt).
#pragma once

namespace caster {

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
// Two cells next to each other in x: GetDensity (medium.h:124-129) of (x0, y, z) and (x1, y, z).  The eight lookups
// of a trilinear sample are four such pairs, and the two cells of a pair are neighbours in memory: where both lie
// inside the grid they come with one 8-byte load (the vector cache's rate of lane addresses is what a tracking loop
// waits for, so four addresses per sample instead of eight).
__device__ inline void het_cell_pair(const DevScene& S, const GridRec& g, int x0, int x1, int y, int z, float& a, float& b) {
  a = 0.f; b = 0.f;
  if (y < 0 || y > g.ny - 1 || z < 0 || z > g.nz - 1) return;
  const bool in0 = x0 >= 0 && x0 <= g.nx - 1, in1 = x1 >= 0 && x1 <= g.nx - 1;
  const float* row = S.density + ((size_t)g.off + (size_t)z * g.ny * g.nx + (size_t)y * g.nx);
  if (in0 && in1 && x1 == x0 + 1) {
    float2 v;
    __builtin_memcpy(&v, row + x0, sizeof(v));   // (4-byte aligned)
    a = v.x; b = v.y;
  } else {
    if (in0) a = row[x0];
    if (in1) b = row[x1];
  }
}
__device__ inline float het_density(const DevScene& S, const GridRec& g, V3 p) {  // medium.h:111-122
  V3 ps = v3(p.x * g.nx - .5f, p.y * g.ny - .5f, p.z * g.nz - .5f);
  V3 psi = v3(floorf(ps.x), floorf(ps.y), floorf(ps.z));
  V3 delta = ps - psi;
  // (the reference adds 1 to the floored coordinate as a float and converts each corner by itself: so here)
  const int x0 = fps::f2i_(psi.x), x1 = fps::f2i_(psi.x + 1.f);
  const int y0 = fps::f2i_(psi.y), y1 = fps::f2i_(psi.y + 1.f);
  const int z0 = fps::f2i_(psi.z), z1 = fps::f2i_(psi.z + 1.f);
  float c000, c100, c010, c110, c001, c101, c011, c111;
  het_cell_pair(S, g, x0, x1, y0, z0, c000, c100);
  het_cell_pair(S, g, x0, x1, y1, z0, c010, c110);
  het_cell_pair(S, g, x0, x1, y0, z1, c001, c101);
  het_cell_pair(S, g, x0, x1, y1, z1, c011, c111);
  float d00 = lerp_(c000, c100, delta.x);
  float d10 = lerp_(c010, c110, delta.x);
  float d01 = lerp_(c001, c101, delta.x);
  float d11 = lerp_(c011, c111, delta.x);
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
  const uint32_t sbase = blockIdx.x * stage.region_cap;
  uint32_t n_next = 0, n_shadow = 0, n_probe = 0;
  CASTER_FOR_ROUNDS_CONTIGUOUS(r, count) {
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

}  // namespace caster
