// Part of the gfx950 wavefront renderer (included by ../kernels.hip, in this order: common.h, traverse.h, trace.h, shading.h, bsdf.h, path.h, volpath.h, lighttrace.h, bdpt.h, volpath_queues.h, film.h).
// The volumetric integrator with shadow and probe rays on queues.
#pragma once

namespace caster {

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
  const uint32_t sbase = blockIdx.x * stage.region_cap;
  uint32_t n_next = 0, n_shadow = 0, n_probe = 0;
  CASTER_FOR_ROUNDS_VPTQ(r, count) {
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
  const uint32_t sbase = blockIdx.x * stage.region_cap;
  uint32_t n_next = 0;
  CASTER_FOR_ROUNDS_VPTQ(r, count) {
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

}  // namespace caster
