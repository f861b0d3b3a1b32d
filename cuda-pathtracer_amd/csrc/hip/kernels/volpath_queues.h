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

// What kind of event the queue entry `idx` will be, for grouping (a wrong answer costs divergence, never
// correctness: the vertex below decides everything again for itself): a material type 0..5; kVptScatter, a
// scattering event in the medium before the surface; kVptLight, the events that cost next to nothing (an emitter
// seen by the camera or through a specular chain, an index-matched boundary crossed, a ray that left the scene
// under an environment light); -1, nothing to do.  A ray that left the scene otherwise ends its path with no
// vertex: what the path still owes from its previous vertex is added by k_finish.
constexpr int kVptScatter = 6, kVptLight = 7, kVptLists = 8;
__device__ inline int vptq_event_class(const DevScene& S, const PathState& P, const RayQueue& cur, uint32_t idx, bool env) {
  const float4 h4 = cur.hit[idx];
  const int prim = __float_as_int(h4.y);
  if (prim < 0) return env && cur.d[idx].w >= 0.f ? kVptLight : -1;   // (tmax < 0: a slot outside the image)
  const int slot = cur.slot[idx];
  const uint32_t flags = __float_as_uint(P.li_flags[slot].w);
  const int rmed = (int)((flags & kFMediumMask) >> kFMediumShift) - 1;
  if (rmed != -1) {
    // the free-flight distance Homogeneous::Sample is going to draw (medium.h:18-25)
    uint32_t rng = __float_as_uint(P.beta_rng[slot].w);
    MediumRec med;
    load_medium(S, rmed, med);
    int channel = fps::f2i_(rng_uniform(rng) * 3);
    channel = channel == 3 ? 2 : channel;
    const float sT = channel == 0 ? med.sigmaT.x : (channel == 1 ? med.sigmaT.y : med.sigmaT.z);
    if (-fps::log_(1 - rng_uniform(rng)) / sT < h4.x) return kVptScatter;
  }
  const float4 ids = S.shade[(size_t)prim * kShadeF4 + 6];
  const bool sees_emitters = (flags & kFBounceMask) == 0u || (flags & kFSpecular) != 0u;
  if (__float_as_int(ids.x) == -1 || (sees_emitters && __float_as_int(ids.y) != -1)) return kVptLight;
  const int t = __float_as_int(S.shade[(size_t)prim * kShadeF4 + 7].y);
  return t >= 0 && t < kVptScatter ? t : kVptLight;
}

// One event of Volpath() (pathtracer.cu:1040-1223) at the queue entries `idx` of the lanes with `active` set.
__device__ inline void vptq_events(const LaunchCtx& L, const RayQueue& cur, const StageArea& stage, StageCursor& sc, uint32_t idx,
                                   bool active) {
  const DevScene& S = L.S;
  const PathState& P = L.P;
  bool want_next = false, want_shadow = false, want_probe = false;
  int slot = 0;
  V3 no = v3(0, 0, 0), nd = v3(0, 0, 1), so = no, sd = nd, po = no, pd = nd;
  float stmax = 0.f;
  if (active) {
    slot = cur.slot[idx];
    float4 o4 = cur.o[idx], d4 = cur.d[idx], h4 = cur.hit[idx];
    V3 ro = v3(o4), rd = v3(d4);
    Hit h;
    h.t = h4.x; h.prim = __float_as_int(h4.y); h.b1 = h4.z; h.b2 = h4.w;
    float4 br = P.beta_rng[slot], lf = P.li_flags[slot];
    V3 beta = v3(br), Li = v3(lf);
    uint32_t rng = __float_as_uint(br.w), flags = __float_as_uint(lf.w);
    int bounces = (int)(flags & kFBounceMask);
    bool specular = (flags & kFSpecular) != 0;
    int rmed = (int)((flags & kFMediumMask) >> kFMediumShift) - 1;
    if (flags & kFPending) Li = resolve_pending_vpt(P, slot, Li, flags);
    uint32_t pending = 0;
    int nmed = rmed;
    bool alive = true;
    bool count_bounce = true;
    if (h.prim < 0) {
      if ((bounces == 0 || specular) && S.inf.valid) Li = Li + beta * inf_le(S.inf, rd);
      alive = false;
    } else {
      const float rtmax = h.t;  // the reference's r.tmax after Intersect
      float sampledDist = 0.f;
      bool sampledMedium = false;
      MediumRec med;
      med.g = 0.f; med.type = 0;
      if (rmed != -1) {
        load_medium(S, rmed, med);
        medium_sample(S, rmed, med, ro, rd, rtmax, rng, beta, sampledDist, sampledMedium);   // (no density grid on this path)
      }
      if (is_black(beta)) {
        alive = false;
      } else if (sampledMedium) {
        // a scattering event: light sample with the phase function as the BSDF, then a new direction
        const V3 samplePos = ro + sampledDist * rd;
        V3 radiance, ldir;
        float ltmax, lightPdf, choicePdf;
        sample_one_light(S, rng, samplePos, radiance, ldir, ltmax, lightPdf, choicePdf);
        if (!is_black(radiance)) {
          want_shadow = true;
          so = samplePos; sd = ldir; stmax = ltmax;
          P.pend_x[slot] = f4(beta, phase_eval(med, -rd, ldir));
          P.pend_z[slot] = f4(radiance, lightPdf * choicePdf);
          P.pend_w[slot] = f4(v3(1.f, 1.f, 1.f), (uint32_t)rmed);
          pending = kFPending | kFPendingM;
        }
        const float ux = rng_uniform(rng);
        const float uy = rng_uniform(rng);
        no = samplePos; nd = phase_sample(med, ux, uy); nmed = rmed;
        specular = false;
      } else {
      // (the intersection record is only built for the events that happen at the surface: in a dense medium most
      // events are scattering events, which never look at it)
      Isect is;
      load_isect(S, ro, rd, h, is);
      const V3 pos = is.pos, nor = is.n;
      if ((bounces == 0 || specular) && is.light != -1) {
        V3 tr = v3(1.f, 1.f, 1.f);
        if (rmed != -1) tr = vexp(med.sigmaT * (-rtmax));
        LightRec l;
        load_light(S, is.light, l);
        V3 le = dot(nor, -rd) > 0.f ? l.radiance : v3(0, 0, 0);
        Li = Li + tr * beta * le;
        alive = false;
      } else if (is.mat == -1) {
        // (the number of boundaries crossed so far rides in beta_prev.w, which only this event and a vertex
        // that leaves a term pending touch)
        const uint32_t crossings = __float_as_uint(P.beta_prev[slot].w) + 1u;
        if (crossings > (uint32_t)kMaxCrossings) alive = false;
        P.beta_prev[slot] = make_float4(0, 0, 0, __uint_as_float(crossings));
        count_bounce = false;
        nmed = dot(rd, is.n) > 0 ? is.med_out : is.med_in;
        no = pos; nd = rd;
      } else {
        Mat m;
        load_mat(S, is.mat, m);
        const bool delta = is_delta(m.type);
        // every random number of the vertex, in the reference's order: light choice and point (3), light
        // probe (3), continuation (3) -- a specular surface draws the last three only
        V3 radiance = v3(0, 0, 0), ldir = v3(0, 0, 1), us = v3(0, 0, 0), u;
        float ltmax = 0.f, lightPdf = 1.f, choicePdf = 1.f;
        if (!delta) {
          sample_one_light(S, rng, pos, radiance, ldir, ltmax, lightPdf, choicePdf);
          us.x = rng_uniform(rng); us.y = rng_uniform(rng); us.z = rng_uniform(rng);
        }
        u.x = rng_uniform(rng); u.y = rng_uniform(rng); u.z = rng_uniform(rng);
        // the material is asked once: the light's direction, a direction for the light probe, the continuation
        const bool lit = !is_black(radiance);
        V3 lfr = v3(0, 0, 0), mo = v3(0, 0, 1), mfr = v3(0, 0, 0), wo2 = v3(0, 0, 1), fr2 = v3(0, 0, 0);
        float lpdf = 0.f, mpdf = 0.f, pdf2 = 0.f;
        with_lobe(S, m, Surface{-rd, nor, is.dpdu}, is.uv, [&](const auto& lobe) {
          if (!lobe.kDelta) {
            if (lit) lobe.eval(ldir, lfr, lpdf);
            lobe.sample(us, true, mo, mfr, mpdf);
          }
          lobe.sample(u, true, wo2, fr2, pdf2);
        });
        if (!delta) {
          float cosl = 0.f, absdot = 0.f;
          if (lit) {
            const float weight = power_heuristic(lightPdf * choicePdf, lpdf);
            want_shadow = true;
            so = pos; sd = ldir; stmax = ltmax;
            cosl = fabsf(dot(nor, ldir));
            P.pend_x[slot] = f4(lfr, weight);
            P.pend_z[slot] = f4(radiance, lightPdf * choicePdf);
            P.pend_w[slot] = f4(v3(1.f, 1.f, 1.f), (uint32_t)rmed);
            pending |= kFPendingX;
          }
          if (!is_black(mfr) && mpdf != 0) {
            // The probe contributes when its closest hit is an emitter, or the environment when it leaves the
            // scene: without an environment, a ray that misses the padded bounds of all emitters is not sent
            // (its term counts as the zero it would have come to).
            float tn;
            const bool reach = S.inf.valid || (S.n_lights > 0 && box_test(v3(S.emit_lo), v3(S.emit_hi), slab_ray(pos, mo), INFINITY,
                                                                         S.box_scale, tn));
            if (reach) {
              want_probe = true;
              po = pos; pd = mo;
              absdot = fabsf(dot(mo, nor));
              P.pend_y[slot] = f4(mfr, mpdf);
              pending |= kFPendingY;
            }
          }
          // Li += beta * Ld: with neither part present that adds beta * 0, which only
          // matters for a non-finite beta (the sample then turns NaN and is dropped)
          const bool finite_beta = fabsf(beta.x) < INFINITY && fabsf(beta.y) < INFINITY && fabsf(beta.z) < INFINITY;
          if (pending || !finite_beta) {
            pending |= kFPending;
            P.pend_v[slot] = make_float4(absdot, cosl, __int_as_float(rmed), 0.f);
            // the pending term keeps the throughput the vertex was reached with
            P.beta_prev[slot] = f4(beta, __float_as_uint(P.beta_prev[slot].w));
          }
        }
        if (is_black(fr2)) {
          alive = false;
        } else {
          beta = beta * (fr2 * fabsf(dot(nor, wo2)) / pdf2);
          specular = delta;
          int mm = dot(wo2, nor) > 0 ? is.med_out : is.med_in;
          mm = dot(-rd, nor) * dot(wo2, nor) > 0 ? rmed : mm;
          nmed = mm;
          no = pos; nd = wo2;
        }
      }
      }
      if (alive && count_bounce && bounces > 3) {
        float illumate = fps::clamp_(1.f - luminance(beta), 0.f, 1.f);
        if (rng_uniform(rng) < illumate) alive = false;
        else beta = beta / (1 - illumate);
      }
      if (count_bounce) {
        if (bounces + 1 >= S.maxDepth) alive = false;
        bounces += 1;
      }
    }
    want_next = alive;
    flags = (uint32_t)bounces | (specular ? kFSpecular : 0u) | ((uint32_t)(nmed + 1) << kFMediumShift) | pending;
    P.beta_rng[slot] = f4(beta, rng);
    P.li_flags[slot] = f4(Li, flags);
  }
  stage_push(stage.next, sc.base, sc.n_next, want_next, slot, no, -1, nd, INFINITY);
  stage_push(stage.shadow, sc.base, sc.n_shadow, want_shadow, slot, so, -1, sd, stmax);
  stage_push(stage.probe, sc.base, sc.n_probe, want_probe, slot, po, -1, pd, INFINITY);
}

// The shade stage of the volumetric integrator on queues.  As the path tracer's (k_shade_pt): a wavefront sorts the
// entries of a round into short lists in LDS by the kind of event they will be, and runs the event code for 64
// entries of one kind at a time -- a scattering event in the medium, one material's vertex, or the cheap events --
// so that the lanes of a group run the same code; rays that left the scene take no lane at all.
// (Sorted by kind -- CASTER_VPTQ_SORT 1 -- the Cornell box in fog runs at 432 Msamples/s, with one list for whatever
// needs an event at 500, against 448 for a lane per queue entry: the events of this integrator are cheap next to
// the sixteen path-state records a vertex gathers by slot, and a group picked from a wider window of the queue
// touches more lines of each; what pays is that the rays that left the scene take no lane.)
#ifndef CASTER_VPTQ_SORT
#define CASTER_VPTQ_SORT 0
#endif
__global__ __launch_bounds__(kWave, CASTER_SHADE_WAVES) void k_shade_vptq(LaunchCtx L, RayQueue cur, ShadeQueues out, StageArea stage) {
  constexpr int kLists = CASTER_VPTQ_SORT ? kVptLists : 1;
  __shared__ uint32_t s_list[kLists][kShadeListCap];
  const DevScene& S = L.S;
  const uint32_t count = *cur.count;
  const uint32_t lane = threadIdx.x;
  StageCursor sc{blockIdx.x * stage.region_cap, blockIdx.x * stage.region_cap, 0u, 0u, 0u};
  uint32_t q_head = 0, q_len = 0;   // ring buffers: lane c keeps the start and the length of list c
  const bool env = S.inf.valid != 0;
  auto take_group = [&](int c, uint32_t n) {   // the first n (<= 64) entries of list c, one per lane
    uint32_t head = (uint32_t)__builtin_amdgcn_readlane((int)q_head, c);
    uint32_t my = s_list[c][(head + lane) & (kShadeListCap - 1)];
    if ((int)lane == c) { q_head = (q_head + n) & (kShadeListCap - 1); q_len -= n; }
    return my;
  };
  CASTER_FOR_ROUNDS_VPTQ(r, count) {
    const uint32_t idx = r * kWave + lane;
    int cls = -1;
    if (idx < count) {
      if (CASTER_VPTQ_SORT) cls = vptq_event_class(S, L.P, cur, idx, env);
      else cls = __float_as_int(cur.hit[idx].y) >= 0 || (env && cur.d[idx].w >= 0.f) ? 0 : -1;
    }
    unsigned long long todo = lane_ballot(cls >= 0);
    while (todo) {
      int first = __builtin_ctzll(todo);
      int c = __builtin_amdgcn_readlane(cls, first);
      unsigned long long mine = lane_ballot(cls == c);
      uint32_t head = (uint32_t)__builtin_amdgcn_readlane((int)q_head, c);
      uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)q_len, c);
      if (cls == c)
        s_list[c][(head + len + (uint32_t)__popcll(mine & ((1ull << lane) - 1ull))) & (kShadeListCap - 1)] = idx;
      if ((int)lane == c) q_len += (uint32_t)__popcll(mine);
      todo &= ~mine;
    }
    __syncthreads();
    unsigned long long full = lane_ballot(lane < (uint32_t)kLists && q_len >= (uint32_t)kWave);
    while (full) {
      int c = __builtin_ctzll(full);
      uint32_t my = take_group(c, kWave);
      vptq_events(L, cur, stage, sc, my, true);
      full = lane_ballot(lane < (uint32_t)kLists && q_len >= (uint32_t)kWave);
    }
    __syncthreads();
  }
  // what is left in the lists
  for (int c = 0; c < kLists; ++c) {
    uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)q_len, c);
    if (n == 0) continue;
    uint32_t my = take_group(c, n);
    vptq_events(L, cur, stage, sc, my, lane < n);
  }
  __syncthreads();  // the copies below read what other lanes of this wavefront staged
  stage_flush(stage.next, sc.base, sc.n_next, out.next);
  stage_flush(stage.shadow, sc.base, sc.n_shadow, out.shadow);
  stage_flush(stage.probe, sc.base, sc.n_probe, out.probe);
}

// One segment of Tr() (pathtracer.cu:307-331) for every shadow ray whose closest
// hit has just been found: an opaque hit ends the ray with zero transmittance,
// otherwise the segment's medium attenuates it and, at an index-matched boundary,
// the ray goes on from the hit in the medium on the other side.
// HET: some medium of the scene is a density grid, whose transmittance (ratio tracking, medium.h:53-79) draws random
// numbers: they come from the path's own generator, whose state waits in the path's state while the shadow ray is
// under way (volpath_hetero.h), one segment after the other as Tr() walks them.
template <bool HET>
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
        // (an opaque hit ends the ray whatever its normal: the material index is looked at first)
        opaque = __float_as_int(S.shade[(size_t)h.prim * kShadeF4 + 6].x) != -1;
        if (!opaque) {
          Isect is;
          load_isect(S, o, d, h, is);
          med_in = is.med_in; med_out = is.med_out; n = is.n;
        }
      }
      if (opaque) {
        tr = v3(0, 0, 0);
      } else {
        float ray_tmax = found ? h.t : d4.w;
        if (medium != -1) {
          if (HET) {
            float4 br = P.beta_rng[slot];
            uint32_t rng = __float_as_uint(br.w);
            tr = tr * medium_tr(S, medium, o, d, ray_tmax, rng);
            br.w = __uint_as_float(rng);
            P.beta_rng[slot] = br;
          } else {
            MediumRec m;
            load_medium(S, medium, m);
            tr = tr * vexp(m.sigmaT * (-ray_tmax));
          }
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
template <bool HET>
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
    float4 br = make_float4(0, 0, 0, 0);
    uint32_t rng = 1u;
    if (HET) { br = P.beta_rng[slot]; rng = __float_as_uint(br.w); }
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
        if (HET) {
          tr = tr * medium_tr(S, medium, o, d, ray_tmax, rng);
        } else {
          MediumRec m;
          load_medium(S, medium, m);
          tr = tr * vexp(m.sigmaT * (-ray_tmax));
        }
      }
      if (!found) break;
      medium = dot(d, n) > 0 ? med_out : med_in;
      seg_tmax = seg_tmax - ray_tmax;
      o = o + ray_tmax * d;
    }
    if (HET) { br.w = __uint_as_float(rng); P.beta_rng[slot] = br; }
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
