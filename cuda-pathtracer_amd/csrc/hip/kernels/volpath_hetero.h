// Part of the gfx950 wavefront renderer (included by ../kernels.hip after volpath_queues.h).
// The volumetric integrator on queues for scenes in which a medium is a density grid.
#pragma once

namespace caster {

// With a density grid in the scene a transmittance walk draws random numbers (ratio tracking, medium.h:53-79) from
// the path's own generator, in the middle of the vertex's draws: Volpath() (pathtracer.cu:1040-1223) draws
//   [free flight through the ray's medium]  light sample (3)  [Tr() of the shadow ray]
//   scattering event:  new direction (2)                                  Russian roulette (1)
//   surface vertex:    light probe (3)  [Tr() of the probe's one segment]  continuation (3)  Russian roulette (1)
// so the vertex cannot draw "all its numbers first" as the queued vertex of a scene without grids does
// (volpath_queues.h).  Here the vertex is cut at the two places where a ray has to be traced before the next number
// can be drawn, and the generator's state waits in the path's state meanwhile:
//   phase A  (k_vpth<0>, over the traced extension queue)  up to the light sample; the shadow ray goes to the Tr()
//            queue, which k_tr_step<true> advances a segment per closest-hit pass, drawing the segment's numbers from
//            the path's generator (one ray per path and round, segments in order: the stream is the reference's);
//   phase B  (k_vpth<1>, over the same queue, the paths that wait for their Tr())  the new direction of a scattering
//            event, or the probe direction of a surface vertex; the probe ray goes through k_trace;
//   phase C  (k_vpth<2>, over the traced probe queue)  the probe's term, the continuation.
// A phase goes on into the next when it has nothing to wait for (no shadow ray, no probe).  Every loop over a
// medium (delta tracking in A, ratio tracking in k_tr_step and C) is then run by all lanes of a wavefront together,
// by kernels a third the size of the one-kernel vertex (k_shade_vpt, which walked every ray in place, lane by lane,
// at two wavefronts per SIMD and 16 % lane utilisation).  What a vertex leaves pending -- the operands of its
// light-sampled terms -- is what the queued vertex leaves, and the next vertex (or k_finish) evaluates it the same way.

constexpr uint32_t kFStageShift = 29;               // li_flags bits 29..30: what a path waits for between the phases
constexpr uint32_t kFStageMask = 3u << kFStageShift;
enum : uint32_t { kStageNone = 0u, kStageTrScatter = 1u, kStageTrSurface = 2u, kStageProbe = 3u };

// ---- tracking loops with lanes that refill ---------------------------------------------------------------------------
// Delta tracking (the free flight through a density grid, medium.h:81-109) and ratio tracking (its transmittance,
// medium.h:53-79) are loops whose trip count differs from ray to ray like a geometric distribution: with one queue
// entry per lane and round, a wavefront runs as long as its longest loop and a quarter to a third of its lanes work.
// The two kernels below give a lane its next queue entry as soon as it has finished one (as the traversal kernel
// refills its lanes): every iteration of the wavefront's loop is one tracking step for all lanes that are in a loop,
// then the lanes whose loop has ended write their result, and once enough lanes are free they take the next entries of
// the wavefront's part of the queue.  Each entry still runs the reference's sequence of operations and draws.
struct LaneFeed {
  uint32_t next, end;   // the wavefront's part of the queue, the same in every lane
};
#ifndef CASTER_FEED_IDLE
#define CASTER_FEED_IDLE 32
#endif
// The tracking kernels are slim (64-73 registers) and wait for memory most of the time: they are launched with this
// multiple of the shade grid (which is sized for the vertex kernels' register count), up to 8 wavefronts per SIMD.
#ifndef CASTER_TRACK_GRID_MUL
#define CASTER_TRACK_GRID_MUL 2
#endif
constexpr uint32_t kTrackGridMul = CASTER_TRACK_GRID_MUL;   // (at most 8: a region holds 8 rounds more than an even share)
constexpr int kFeedIdle = CASTER_FEED_IDLE;   // lanes free before the next entries are taken
// Gives the free lanes the next entries: this lane's entry, or ~0u.
__device__ inline uint32_t feed_take(LaneFeed& f, bool idle) {
  const unsigned long long m = lane_ballot(idle);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t n_idle = (uint32_t)__popcll(m), avail = f.end - f.next, give = avail < n_idle ? avail : n_idle;
  const uint32_t prefix = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  const uint32_t mine = (idle && prefix < give) ? f.next + prefix : ~0u;
  f.next += give;
  return mine;
}
__device__ inline LaneFeed feed_of(uint32_t count) {   // a contiguous part per wavefront, in whole rounds of 64
  const uint32_t all = (count + kWave - 1) / kWave, per = (all + gridDim.x - 1) / gridDim.x;
  const uint32_t r0 = blockIdx.x * per, r1 = r0 + per < all ? r0 + per : all;
  LaneFeed f;
  f.next = r0 < all ? r0 * kWave : count;
  f.end = r1 * kWave < count ? r1 * kWave : count;
  if (f.next > f.end) f.next = f.end;
  return f;
}
// What a tracking loop keeps of its medium.
struct TrackMedium {
  GridRec g;
  V3 ext;
  float sigma, invMax;
};
__device__ inline void track_medium(const DevScene& S, int medium, const MediumRec& m, TrackMedium& t) {
  load_grid(S, medium, t.g);
  t.sigma = dot(m.sigmaT, v3(0.212671f, 0.715160f, 0.072169f));
  t.ext = t.g.p1 - t.g.p0;
  t.invMax = m.invMax;
}

// Medium::Sample (medium.h:18-38, :81-109) for every entry of the traced extension queue whose ray runs through a
// medium and hit something: scales the path's beta, advances its generator, and leaves the distance of a scattering
// event before the surface in the entry's o.w (which holds -1, "no hit known beforehand", otherwise).
__global__ __launch_bounds__(kWave) void k_vpth_flight(LaunchCtx L, RayQueue cur) {
  const DevScene& S = L.S;
  const PathState& P = L.P;
  LaneFeed feed = feed_of(*cur.count);
  bool busy = false, looping = false;
  uint32_t idx = 0, rng = 1u;
  int slot = 0, iter = 0;
  V3 ro = v3(0, 0, 0), rd = v3(0, 0, 1), beta = v3(0, 0, 0), scatter_scale = v3(0, 0, 0);
  float rtmax = 0.f, dist = 0.f, sampledDist = 0.f;
  bool sampledMedium = false;
  TrackMedium tm;
  tm.sigma = 1.f; tm.invMax = 1.f; tm.ext = v3(1, 1, 1);
  tm.g.nx = tm.g.ny = tm.g.nz = 1; tm.g.off = 0; tm.g.iterMax = 1; tm.g.p0 = tm.g.p1 = v3(0, 0, 0);
  for (;;) {
    const int n_busy = __popcll(lane_ballot(busy));
    if (feed.next < feed.end && kWave - n_busy >= kFeedIdle) {
      const uint32_t mine = feed_take(feed, !busy);
      if (mine != ~0u) {
        // (the loads are written out level by level, so that a refill waits for three round trips to memory, not six)
        const float4 h4 = cur.hit[mine], o4 = cur.o[mine], d4 = cur.d[mine];
        const int s = cur.slot[mine];
        const uint32_t flags = __float_as_uint(P.li_flags[s].w);
        const float4 br = P.beta_rng[s];
        const int rmed = (int)((flags & kFMediumMask) >> kFMediumShift) - 1;
        if (__float_as_int(h4.y) >= 0 && rmed != -1) {
          idx = mine; slot = s;
          ro = v3(o4); rd = v3(d4); rtmax = h4.x;
          beta = v3(br); rng = __float_as_uint(br.w);
          MediumRec med;
          load_medium(S, rmed, med);
          track_medium(S, rmed, med, tm);   // (the record of a homogeneous medium has these fields too, unused)
          busy = true;
          sampledMedium = false; sampledDist = 0.f;
          if (med.type == 0) {
            medium_sample(S, rmed, med, ro, rd, rtmax, rng, beta, sampledDist, sampledMedium);   // (no loop in there)
            looping = false;
          } else {
            scatter_scale = med.sigmaS / med.sigmaT;
            dist = 0.f; iter = tm.g.iterMax;
            looping = true;
          }
        }
      }
    } else if (n_busy == 0) {
      break;   // (nothing under way and nothing left to take)
    }
    if (busy && looping) {   // one step of delta tracking (medium.h:88-104)
      dist += -fps::log_(rng_uniform(rng)) * tm.invMax / tm.sigma;
      if (dist >= rtmax) {
        beta = beta * v3(1.f, 1.f, 1.f);
        looping = false;
      } else {
        V3 p = ro + rd * dist;
        p = (p - tm.g.p0) / tm.ext;
        if (het_density(S, tm.g, p) * tm.invMax > rng_uniform(rng)) {
          sampledDist = dist;
          sampledMedium = true;
          beta = beta * scatter_scale;
          looping = false;
        } else if (--iter == 0) {
          beta = beta * v3(1.f, 1.f, 1.f);
          looping = false;
        }
      }
    }
    if (busy && !looping) {
      P.beta_rng[slot] = f4(beta, rng);
      if (sampledMedium) cur.o[idx].w = sampledDist;
      busy = false;
    }
  }
}

// One segment of Tr() for every shadow ray whose closest hit has just been found, where a medium may be a density
// grid: k_tr_step<true> with lanes that refill.
__global__ __launch_bounds__(kWave) void k_tr_step_grid(LaunchCtx L, RayQueue cur, RayQueue next, StageArea stage) {
  const DevScene& S = L.S;
  const PathState& P = L.P;
  LaneFeed feed = feed_of(*cur.count);
  // (launched with kTrackGridMul times the shade grid the staging regions are sized for: each wavefront's share of the
  // queue, and with it the part of a region it can fill, is that much smaller)
  const uint32_t sbase = blockIdx.x * (stage.region_cap / kTrackGridMul);
  uint32_t n_next = 0;
  bool busy = false, looping = false, found = false;
  uint32_t rng = 1u;
  int slot = 0, iter = 0, medium = -1, med_in = -1, med_out = -1;
  V3 o = v3(0, 0, 0), d = v3(0, 0, 1), n = v3(0, 0, 0), tr = v3(0, 0, 0);
  float seg_tmax = 0.f, ray_tmax = 0.f, dist = 0.f, trs = 1.f;
  bool draws = false;
  TrackMedium tm;
  tm.sigma = 1.f; tm.invMax = 1.f; tm.ext = v3(1, 1, 1);
  tm.g.nx = tm.g.ny = tm.g.nz = 1; tm.g.off = 0; tm.g.iterMax = 1; tm.g.p0 = tm.g.p1 = v3(0, 0, 0);
  for (;;) {
    const int n_busy = __popcll(lane_ballot(busy));
    if (feed.next < feed.end && kWave - n_busy >= kFeedIdle) {
      const uint32_t mine = feed_take(feed, !busy);
      if (mine != ~0u) {
        slot = cur.slot[mine];
        const float4 o4 = cur.o[mine], d4 = cur.d[mine], h4 = cur.hit[mine];
        o = v3(o4); d = v3(d4); seg_tmax = d4.w;
        Hit h;
        h.t = h4.x; h.prim = __float_as_int(h4.y); h.b1 = h4.z; h.b2 = h4.w;
        const float4 w4 = P.pend_w[slot];
        const uint32_t rng_at = __float_as_uint(P.beta_rng[slot].w);
        tr = v3(w4);
        medium = __float_as_int(w4.w);
        MediumRec m;
        m.type = 0; m.sigmaT = v3(0, 0, 0);
        if (medium != -1) {   // (asked for before the hit is looked at: one round trip to memory less per refill)
          load_medium(S, medium, m);
          track_medium(S, medium, m, tm);
        }
        found = h.prim >= 0;
        med_in = -1; med_out = -1; n = v3(0, 0, 0);
        bool opaque = false;
        if (found) {
          opaque = __float_as_int(S.shade[(size_t)h.prim * kShadeF4 + 6].x) != -1;
          if (!opaque) {
            Isect is;
            load_isect(S, o, d, h, is);
            med_in = is.med_in; med_out = is.med_out; n = is.n;
          }
        }
        busy = true; looping = false; draws = false;
        if (opaque) {
          tr = v3(0, 0, 0);
          found = false;   // (the ray ends here)
        } else {
          ray_tmax = found ? h.t : seg_tmax;
          if (medium != -1) {
            if (m.type == 0) {
              tr = tr * vexp(m.sigmaT * (-ray_tmax));
            } else {   // ratio tracking (medium.h:53-79), a step per iteration of the loop below
              rng = rng_at;
              trs = 1.f; dist = 0.f; iter = tm.g.iterMax;
              looping = true; draws = true;
            }
          }
        }
      }
    } else if (n_busy == 0) {
      break;
    }
    bool want = false;
    float remaining = 0.f;
    if (busy && looping) {
      dist += -fps::log_(rng_uniform(rng)) * tm.invMax / tm.sigma;
      bool done = dist >= ray_tmax, zero = false;
      if (!done) {
        V3 p = o + d * dist;
        p = (p - tm.g.p0) / tm.ext;
        trs *= 1.f - het_density(S, tm.g, p) * tm.invMax;
        if (trs < 0.1f) {
          const float q = 1.f - trs;
          if (rng_uniform(rng) < q) { done = true; zero = true; }
          else trs /= (1.f - q);
        }
        if (!done && --iter == 0) done = true;
      }
      if (done) {
        tr = tr * (zero ? v3(0.f, 0.f, 0.f) : v3(trs, trs, trs));
        looping = false;
      }
    }
    if (busy && !looping) {
      if (found) {
        medium = dot(d, n) > 0 ? med_out : med_in;
        remaining = seg_tmax - ray_tmax;
        o = o + ray_tmax * d;
        want = true;
      }
      P.pend_w[slot] = f4(tr, (uint32_t)medium);
      if (draws) P.beta_rng[slot].w = __uint_as_float(rng);
      busy = false;
    }
    stage_push(stage.shadow, sbase, n_next, want, slot, o, -1, d, remaining);
  }
  __syncthreads();
  stage_flush(stage.shadow, sbase, n_next, next);
}

template <int PHASE>
__global__ __launch_bounds__(kWave, CASTER_SHADE_WAVES) void k_vpth(LaunchCtx L, RayQueue cur, RayQueue probe_in, ShadeQueues out,
                                                                    StageArea stage) {
  const DevScene& S = L.S;
  const PathState& P = L.P;
  const RayQueue& work = PHASE == 2 ? probe_in : cur;
  const uint32_t count = *work.count;
  StageCursor sc{blockIdx.x * stage.region_cap, blockIdx.x * stage.region_cap, 0u, 0u, 0u};
  CASTER_FOR_ROUNDS_CONTIGUOUS(r, count) {
    const uint32_t idx = r * kWave + threadIdx.x;
    bool want_next = false, want_shadow = false, want_probe = false;
    int slot = 0;
    V3 no = v3(0, 0, 0), nd = v3(0, 0, 1), so = no, sd = nd, po = no, pd = nd;
    float stmax = 0.f;
    // which path, which entry of the extension queue, and whether this phase has something to do for it
    uint32_t cidx = idx;
    bool active = idx < count;
    uint32_t flags = 0;
    float4 lf = make_float4(0, 0, 0, 0);
    if (active) {
      slot = work.slot[idx];
      if (PHASE == 2) cidx = __float_as_uint(P.pend_v[slot].w);
      lf = P.li_flags[slot];
      flags = __float_as_uint(lf.w);
      const uint32_t waits = (flags & kFStageMask) >> kFStageShift;
      if (PHASE == 0) active = cur.d[idx].w >= 0.f || __float_as_int(cur.hit[idx].y) >= 0;   // (a slot outside the image)
      if (PHASE == 1) active = waits == kStageTrScatter || waits == kStageTrSurface;
      if (PHASE == 2) active = waits == kStageProbe;
    }
    if (active) {
      const float4 o4 = cur.o[cidx], d4 = cur.d[cidx], h4 = cur.hit[cidx];
      const V3 ro = v3(o4), rd = v3(d4);
      Hit h;
      h.t = h4.x; h.prim = __float_as_int(h4.y); h.b1 = h4.z; h.b2 = h4.w;
      const float4 br = P.beta_rng[slot];
      V3 beta = v3(br), Li = v3(lf);
      uint32_t rng = __float_as_uint(br.w);
      int bounces = (int)(flags & kFBounceMask);
      bool specular = (flags & kFSpecular) != 0;
      const int rmed = (int)((flags & kFMediumMask) >> kFMediumShift) - 1;
      const uint32_t waits = (flags & kFStageMask) >> kFStageShift;
      uint32_t pending = PHASE == 0 ? 0u : (flags & (kFPending | kFPendingM | kFPendingX | kFPendingY));
      int nmed = rmed;
      bool alive = true, count_bounce = true;
      // where the event stands
      enum { kStart, kScatterTail, kAfterLight, kAfterProbe, kContinue, kEnd, kParked };
      int step = PHASE == 0 ? kStart : PHASE == 1 ? (waits == kStageTrScatter ? kScatterTail : kAfterLight) : kAfterProbe;
      uint32_t park = kStageNone;
      // the surface the event happens at (built when first asked for)
      Isect is;
      Mat m;
      bool have_surface = false;
      auto surface = [&]() {
        if (!have_surface) { load_isect(S, ro, rd, h, is); if (is.mat >= 0) load_mat(S, is.mat, m); have_surface = true; }
      };

      if (PHASE == 0 && step == kStart) {
        if (flags & kFPending) Li = resolve_pending_vpt(P, slot, Li, flags);
        if (h.prim < 0) {
          if ((bounces == 0 || specular) && S.inf.valid) Li = Li + beta * inf_le(S.inf, rd);
          alive = false;
          step = kEnd;
        } else {
          const float rtmax = h.t;  // the reference's r.tmax after Intersect
          float sampledDist = 0.f;
          bool sampledMedium = false;
          MediumRec med;
          med.g = 0.f; med.type = 0;
          if (rmed != -1) {
            // the free flight has been sampled by k_vpth_flight: beta and the generator are past it, and the entry's
            // o.w holds the distance of a scattering event before the surface, if there was one
            load_medium(S, rmed, med);
            sampledMedium = __float_as_int(o4.w) != -1;
            sampledDist = o4.w;
          }
          if (is_black(beta)) {
            alive = false;
            step = kEnd;
          } else if (sampledMedium) {
            const V3 samplePos = ro + sampledDist * rd;
            V3 radiance, ldir;
            float ltmax, lightPdf, choicePdf;
            sample_one_light(S, rng, samplePos, radiance, ldir, ltmax, lightPdf, choicePdf);
            // the reference walks Tr() whatever the light sample is (pathtracer.cu:1079), and with a density grid in the
            // scene the walk draws numbers: the shadow ray is always sent
            want_shadow = true;
            so = samplePos; sd = ldir; stmax = ltmax;
            P.pend_w[slot] = f4(v3(1.f, 1.f, 1.f), (uint32_t)rmed);
            if (!is_black(radiance)) {
              P.pend_x[slot] = f4(beta, phase_eval(med, -rd, ldir));
              P.pend_z[slot] = f4(radiance, lightPdf * choicePdf);
              pending = kFPending | kFPendingM;
            }
            P.pend_y[slot] = f4(samplePos, 0.f);   // (where the path goes on from, once its Tr() has been walked)
            park = kStageTrScatter;
            step = kParked;
          } else {
            surface();
            if ((bounces == 0 || specular) && is.light != -1) {
              V3 tr = v3(1.f, 1.f, 1.f);
              if (rmed != -1) tr = medium_tr(S, rmed, ro, rd, rtmax, rng);
              LightRec l;
              load_light(S, is.light, l);
              V3 le = dot(is.n, -rd) > 0.f ? l.radiance : v3(0, 0, 0);
              Li = Li + tr * beta * le;
              alive = false;
              step = kEnd;
            } else if (is.mat == -1) {
              const uint32_t crossings = __float_as_uint(P.beta_prev[slot].w) + 1u;
              if (crossings > (uint32_t)kMaxCrossings) alive = false;
              P.beta_prev[slot] = make_float4(0, 0, 0, __uint_as_float(crossings));
              count_bounce = false;
              nmed = dot(rd, is.n) > 0 ? is.med_out : is.med_in;
              no = is.pos; nd = rd;
              step = kEnd;
            } else if (is_delta(m.type)) {
              step = kContinue;
            } else {
              // Li += beta * Ld is owed whatever Ld turns out to be (a non-finite beta makes the sample invalid even
              // when Ld is zero): the vertex always leaves its throughput behind
              V3 radiance, ldir;
              float ltmax, lightPdf, choicePdf;
              sample_one_light(S, rng, is.pos, radiance, ldir, ltmax, lightPdf, choicePdf);
              pending = kFPending;
              P.beta_prev[slot] = f4(beta, __float_as_uint(P.beta_prev[slot].w));
              float cosl = 0.f;
              step = kAfterLight;
              if (!is_black(radiance)) {
                V3 lfr;
                float lpdf;
                eval_bsdf(S, m, -rd, ldir, is.n, is.uv, is.dpdu, lfr, lpdf);
                const float weight = power_heuristic(lightPdf * choicePdf, lpdf);
                want_shadow = true;
                so = is.pos; sd = ldir; stmax = ltmax;
                cosl = fabsf(dot(is.n, ldir));
                P.pend_x[slot] = f4(lfr, weight);
                P.pend_z[slot] = f4(radiance, lightPdf * choicePdf);
                P.pend_w[slot] = f4(v3(1.f, 1.f, 1.f), (uint32_t)rmed);
                pending |= kFPendingX;
                park = kStageTrSurface;
                step = kParked;
              }
              P.pend_v[slot] = make_float4(0.f, cosl, __int_as_float(rmed), __uint_as_float(cidx));
            }
          }
        }
      }
      if (PHASE <= 1 && step == kScatterTail) {
        // Medium::SamplePhase (medium.h:146-172); the medium's parameters again, the place from the path's state
        MediumRec med;
        load_medium(S, rmed, med);
        const float ux = rng_uniform(rng);
        const float uy = rng_uniform(rng);
        no = v3(P.pend_y[slot]); nd = phase_sample(med, ux, uy); nmed = rmed;
        specular = false;
        step = kEnd;
      }
      if (PHASE <= 1 && step == kAfterLight) {
        surface();
        V3 us;
        us.x = rng_uniform(rng); us.y = rng_uniform(rng); us.z = rng_uniform(rng);
        V3 mo, mfr;
        float mpdf;
        sample_bsdf(S, m, -rd, is.n, is.uv, is.dpdu, us, mo, mfr, mpdf);
        step = kContinue;
        // (a probe that misses the padded bounds of all emitters cannot end on one, and without an environment light
        // it then contributes nothing and draws nothing: it is not sent)
        float tn;
        const bool reach = S.inf.valid || (S.n_lights > 0 && box_test(v3(S.emit_lo), v3(S.emit_hi), slab_ray(is.pos, mo), INFINITY,
                                                                     S.box_scale, tn));
        if (!is_black(mfr) && mpdf != 0 && reach) {
          want_probe = true;
          po = is.pos; pd = mo;
          P.pend_y[slot] = f4(mfr, mpdf);
          float4 pv = P.pend_v[slot];
          pv.x = fabsf(dot(mo, is.n));
          P.pend_v[slot] = pv;
          pending |= kFPendingY;
          park = kStageProbe;
          step = kParked;
        }
      }
      if (PHASE == 2 && step == kAfterProbe) {
        // the BSDF-sampled direct light (pathtracer.cu:1148-1194) of the probe that has just been traced
        const float4 po4 = probe_in.o[idx], pd4 = probe_in.d[idx], ph4 = probe_in.hit[idx];
        const V3 pos = v3(po4), mo = v3(pd4);
        Hit lh;
        lh.t = ph4.x; lh.prim = __float_as_int(ph4.y); lh.b1 = ph4.z; lh.b2 = ph4.w;
        const float4 py = P.pend_y[slot];
        const V3 mfr = v3(py);
        const float mpdf = py.w, absdot = P.pend_v[slot].x;
        V3 y = v3(0, 0, 0);
        if (lh.prim >= 0) {
          int light = __float_as_int(S.shade[(size_t)lh.prim * kShadeF4 + 6].y);
          uint32_t type = __float_as_uint(S.shade[(size_t)lh.prim * kShadeF4 + 7].x) & kPrimTypeMask;
          if (type == 0u && light != -1) {
            LightRec l;
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
              y = weight * tr * mfr * rad2 * absdot / mpdf;
            }
          }
        } else if (S.inf.valid) {
          V3 rad2 = inf_le(S.inf, mo);
          float cp = S.light_cdf[S.n_lights + 1] - S.light_cdf[S.n_lights];
          float weight = power_heuristic(mpdf, ONE_OVER_FOUR_PI * cp);
          V3 tr = v3(1.f, 1.f, 1.f);
          if (rmed != -1) tr = medium_tr(S, rmed, pos, mo, INFINITY, rng);
          y = weight * tr * mfr * rad2 * absdot / mpdf;
        }
        P.pend_y[slot] = f4(y, 0.f);
        step = kContinue;
      }
      if (step == kContinue) {
        surface();
        V3 u;
        u.x = rng_uniform(rng); u.y = rng_uniform(rng); u.z = rng_uniform(rng);
        V3 wo2, fr2;
        float pdf2;
        sample_bsdf(S, m, -rd, is.n, is.uv, is.dpdu, u, wo2, fr2, pdf2);
        if (is_black(fr2)) {
          alive = false;
        } else {
          beta = beta * (fr2 * fabsf(dot(is.n, wo2)) / pdf2);
          specular = is_delta(m.type);
          int mm = dot(wo2, is.n) > 0 ? is.med_out : is.med_in;
          mm = dot(-rd, is.n) * dot(wo2, is.n) > 0 ? rmed : mm;
          nmed = mm;
          no = is.pos; nd = wo2;
        }
        step = kEnd;
      }
      if (step == kEnd) {
        if (alive && count_bounce && bounces > 3) {
          float illumate = fps::clamp_(1.f - luminance(beta), 0.f, 1.f);
          if (rng_uniform(rng) < illumate) alive = false;
          else beta = beta / (1 - illumate);
        }
        if (count_bounce) {
          if (bounces + 1 >= S.maxDepth) alive = false;
          bounces += 1;
        }
        want_next = alive;
        flags = (uint32_t)bounces | (specular ? kFSpecular : 0u) | ((uint32_t)(nmed + 1) << kFMediumShift) | pending;
      } else {   // parked: the event goes on in a later phase, with the ray's medium and the bounce count as they are
        flags = (flags & (kFBounceMask | kFSpecular | kFMediumMask)) | pending | (park << kFStageShift);
      }
      P.beta_rng[slot] = f4(beta, rng);
      P.li_flags[slot] = f4(Li, flags);
    }
    stage_push(stage.next, sc.base, sc.n_next, want_next, slot, no, -1, nd, INFINITY);
    stage_push(stage.shadow, sc.base, sc.n_shadow, want_shadow, slot, so, -1, sd, stmax);
    stage_push(stage.probe, sc.base, sc.n_probe, want_probe, slot, po, -1, pd, INFINITY);
  }
  __syncthreads();  // the copies below read what other lanes of this wavefront staged
  stage_flush(stage.next, sc.base, sc.n_next, out.next);
  if (PHASE == 0) stage_flush(stage.shadow, sc.base, sc.n_shadow, out.shadow);
  if (PHASE <= 1) stage_flush(stage.probe, sc.base, sc.n_probe, out.probe);
}

}  // namespace caster
