// Part of the gfx950 wavefront renderer (included by ../kernels.hip, in this order: common.h, traverse.h, trace.h, shading.h, bsdf.h, path.h, volpath.h, lighttrace.h, bdpt.h, volpath_queues.h, film.h).
// Queue staging, the path-tracing vertex (k_shade_pt), its probe resolution, the AO vertex.
#pragma once

namespace caster {

// Queues a path writes to while being shaded.
struct ShadeQueues {
  RayQueue next;    // continuation rays (closest hit)
  RayQueue shadow;  // light-sample rays (any hit)
  RayQueue probe;   // BSDF-sampled direct-light rays (closest hit + weighting)
};

// Two-level queue compaction.  A shading wavefront appends the rays it emits,
// ballot-compacted, to a staging region of its own (no atomics: the cursor is
// a wave-uniform register), and when it has shaded all its paths it reserves
// its final range in the shared queue with ONE atomic per queue and copies the
// region there.  Appending straight to the shared queue costs one same-address
// atomic per 64 paths per queue, which is what bounded the shade stage before.
struct StageArea {
  RayQueue next, shadow, probe;  // o/d/slot arrays of gridDim.x * region_cap entries; count unused
  uint32_t region_cap;           // entries per wavefront region
  uint32_t shadow_region_cap;    // the same for `shadow` where the path tracer also appends its probe rays to it
};
__device__ inline void stage_push(const RayQueue& stage, uint32_t base, uint32_t& n, bool want, int slot, V3 o,
                                  int seed, V3 d, float tmax) {
  unsigned long long mask = lane_ballot(want);
  if (mask == 0) return;
  uint32_t lane = threadIdx.x & 63u;
  if (want) {
    uint32_t at = base + n + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    stage.o[at] = f4(o, __int_as_float(seed));
    stage.d[at] = f4(d, tmax);
    stage.slot[at] = slot;
  }
  n += (uint32_t)__popcll(mask);
}
__device__ inline void stage_flush(const RayQueue& stage, uint32_t base, uint32_t n, const RayQueue& fin) {
  n = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
  if (n == 0) return;
  uint32_t lane = threadIdx.x & 63u;
  uint32_t off = 0;
  if (lane == 0) off = atomicAdd(fin.count, n);
  off = (uint32_t)__builtin_amdgcn_readfirstlane((int)off);
  for (uint32_t i = lane; i < n; i += kWave) {
    fin.o[off + i] = stage.o[base + i];
    fin.d[off + i] = stage.d[base + i];
    fin.slot[off + i] = stage.slot[base + i];
  }
}
// The rounds (groups of 64 queue entries) wavefront `blockIdx.x` shades: chunks of CASTER_SHADE_CHUNK
// consecutive rounds dealt to the wavefronts round-robin, so that each sees a sample of the whole image.
// (A contiguous range per wavefront made the cost per wavefront follow the image -- sky here, car paint
// there -- and a launch as long as its dearest range: the path tracer's shade launches were 15 % longer.)
#ifndef CASTER_SHADE_CHUNK
#define CASTER_SHADE_CHUNK 8
#endif
// (the kernels that walk the tree in place, lane by lane, keep a contiguous range per wavefront: their cost is the
// walks, and neighbouring rounds walk the same part of the tree -- the volumetric integrator with a density grid ran
// 21 % slower with its rounds dealt round-robin)
#define CASTER_FOR_ROUNDS_CONTIGUOUS(r, count)                                                               \
  for (uint32_t all_ = ((count) + kWave - 1) / kWave, per_ = (all_ + gridDim.x - 1) / gridDim.x,             \
                r = blockIdx.x * per_, end_ = r + per_ < all_ ? r + per_ : all_;                             \
       r < end_; ++r)
// (measured with their rounds dealt round-robin instead: AO on the Cornell box 5979 against 6109 Msamples/s, light
// tracing on the benchmark scene 1256 against 1313; the queued volumetric integrator, since its shade stage skips
// the rays that left the scene, gains from it like the path tracer's: 530 against 493)
#ifndef CASTER_FOR_ROUNDS_AO
#define CASTER_FOR_ROUNDS_AO CASTER_FOR_ROUNDS_CONTIGUOUS
#endif
#ifndef CASTER_FOR_ROUNDS_LT
#define CASTER_FOR_ROUNDS_LT CASTER_FOR_ROUNDS_CONTIGUOUS
#endif
#ifndef CASTER_FOR_ROUNDS_VPTQ
#define CASTER_FOR_ROUNDS_VPTQ CASTER_FOR_ROUNDS
#endif
#define CASTER_FOR_ROUNDS(r, count)                                                                          \
  for (uint32_t c0_ = blockIdx.x * CASTER_SHADE_CHUNK, all_ = ((count) + kWave - 1) / kWave; c0_ < all_;       \
       c0_ += gridDim.x * CASTER_SHADE_CHUNK)                                                                 \
    for (uint32_t r = c0_; r < c0_ + CASTER_SHADE_CHUNK && r < all_; ++r)

// Adds the two direct-light terms left pending by the previous vertex:
// Li += beta * (Ld_light + Ld_bsdf)  (pathtracer.cu:984).
__device__ inline V3 resolve_pending(const PathState& P, int slot, V3 Li, uint32_t flags) {
  V3 x = v3(P.pend_x[slot]);
  V3 y = (flags & kFPendingY) ? v3(P.pend_y[slot]) : v3(0.f, 0.f, 0.f);
  V3 bp = v3(P.beta_prev[slot]);
  V3 Ld = x + y;
  return Li + bp * Ld;
}

// The BSDF-sampled direct-light term of a probe ray that reaches emitter triangle `prim` at
// distance t, barycentrics (b1, b2):  weight * fr * Le * |cos| / pdf  (pathtracer.cu:949-983).
__device__ inline V3 probe_emitter_term(const DevScene& S, int prim, V3 pos, V3 out, float t, float b1, float b2, V3 fr,
                                        float pdf, float absdot) {
  V3 y = v3(0, 0, 0);
  int light = __float_as_int(S.shade[(size_t)prim * kShadeF4 + 6].y);
  uint32_t type = __float_as_uint(S.shade[(size_t)prim * kShadeF4 + 7].x) & kPrimTypeMask;
  if (type == 0u && light != -1) {
    LightRec l;
    load_light(S, light, l);
    const float4* sh = S.shade + (size_t)prim * kShadeF4;
    V3 n = normalize(v3(sh[0]) * (1.f - b1 - b2) + v3(sh[1]) * b1 + v3(sh[2]) * b2);
    V3 p = pos + t * out;
    V3 radiance = dot(n, -out) > 0.f ? l.radiance : v3(0, 0, 0);
    if (!is_black(radiance)) {
      float pdfA = 1.f / l.area;
      float choicePdf = S.light_cdf[light + 1] - S.light_cdf[light];
      float lenSquare = dot(p - pos, p - pos);
      float costheta = fabsf(dot(n, out));
      float lPdf = pdfA * lenSquare / (costheta);
      float weight = power_heuristic(pdf, lPdf * choicePdf);
      y = weight * fr * radiance * absdot / pdf;
    }
  }
  return y;
}

// One vertex of Path() (pathtracer.cu:897-1007) for every path whose
// extension ray has been traced.
#ifndef CASTER_SHADE_WAVES
#define CASTER_SHADE_WAVES 4
#endif
#ifndef CASTER_SHADE_SORT
#define CASTER_SHADE_SORT 1    // 1: vertices are shaded in groups of one material type; 0: in groups of hits of any type
#endif
#ifndef CASTER_PROBE_CULL
#define CASTER_PROBE_CULL 1
#endif
constexpr int kProbeCullLights = 8;  // emitter triangles tested directly before a probe ray is sent

// Where the rays a shading wavefront emits go: its staging regions and how full they are.
struct StageCursor {
  uint32_t base, base_shadow;
  uint32_t n_next, n_shadow, n_probe;
};

// The vertices at queue entries `idx` of the lanes with `active` set, one per lane.
// PROBE_ANY: the probe rays go to the shadow queue as occlusion rays, their terms settled here (see below;
// the host asks for it where every probe knows its emitter and the batch is small enough to gain).
template <bool PROBE_ANY>
__device__ inline void shade_pt_vertices(const LaunchCtx& L, const RayQueue& cur, const StageArea& stage, StageCursor& sc,
                                         uint32_t idx, bool active) {
  const DevScene& S = L.S;
  const PathState& P = L.P;
  int slot = 0;
  bool want_next = false, want_shadow = false, want_probe = false;
  V3 no = v3(0, 0, 0), nd = v3(0, 0, 1), so = no, sd = nd, po = no, pd = nd;
  float stmax = 0.f, ptmax = INFINITY;
  int pseed = -1;
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
    if (flags & kFPending) {
      Li = resolve_pending(P, slot, Li, flags);
      flags &= ~(kFPending | kFPendingY);
    }
    bool alive = true;
    if (h.prim < 0) {
      if ((bounces == 0 || specular) && S.inf.valid) Li = Li + beta * inf_le(S.inf, rd);
      alive = false;
    }
    if (alive) {
      Isect is;
      load_isect(S, ro, rd, h, is);
      V3 pos = is.pos, nor = is.n;
      if ((bounces == 0 || specular) && is.light != -1) {
        // Area::Le (area.h:36-39)
        LightRec l;
        load_light(S, is.light, l);
        V3 le = dot(nor, -rd) > 0.f ? l.radiance : v3(0, 0, 0);
        Li = Li + beta * le;
        alive = false;
      }
      if (alive) {
        Mat m;
        load_mat(S, is.mat, m);
        const bool delta = is_delta(m.type);
        // every random number of the vertex, in the reference's order: light choice and point (3),
        // light probe (3), continuation (3) -- a specular surface draws the last three only
        V3 radiance = v3(0, 0, 0), ldir = v3(0, 0, 1), us = v3(0, 0, 0), u;
        float ltmax = 0.f, lightPdf = 1.f, choicePdf = 1.f;
        if (!delta) {
          sample_one_light(S, rng, pos, radiance, ldir, ltmax, lightPdf, choicePdf);
          us.x = rng_uniform(rng); us.y = rng_uniform(rng); us.z = rng_uniform(rng);
        }
        u.x = rng_uniform(rng); u.y = rng_uniform(rng); u.z = rng_uniform(rng);
        // the three questions to the material: the light's direction, a direction of its own
        // choosing for the light probe, another for the continuation
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
          V3 x = v3(0, 0, 0);
          if (lit) {
            float weight = power_heuristic(lightPdf * choicePdf, lpdf);
            x = weight * lfr * radiance * fabsf(dot(nor, ldir)) / (lightPdf * choicePdf);
            // The reference traces the shadow ray before it evaluates the BSDF
            // (pathtracer.cu:933-940); when the term is exactly zero (light behind
            // the surface, zero BSDF) visibility cannot change Ld, so no ray is sent.
            want_shadow = !is_black(x);
            so = pos; sd = ldir; stmax = ltmax;
          }
          float4 py = make_float4(0, 0, 0, 0);
          float absdot = 0.f;
          if (!is_black(mfr) && mpdf != 0) {
            want_probe = true;
            po = pos; pd = mo;
            py = f4(mfr, mpdf);
            absdot = fabsf(dot(mo, nor));
#if CASTER_PROBE_CULL
            // The probe only contributes when its closest hit is an emitter (no
            // environment light here), so a ray that cannot reach one is not
            // sent.  With few emitters their triangles are tested with the very
            // test the traversal uses: if none is hit, the closest hit cannot be
            // one; otherwise the ray starts out holding the nearest emitter hit,
            // so that the traversal only looks for something at least as close.
            if (!S.inf.valid) {
              // first the padded box around all emitters: a ray that misses it cannot hit one
              float tn;
              bool reach = S.n_lights > 0 &&
                           box_test(v3(S.emit_lo), v3(S.emit_hi), slab_ray(pos, mo), INFINITY, S.box_scale, tn);
              if (!reach) {
                want_probe = false;
                py = make_float4(0, 0, 0, 0);
              } else if (S.n_lights <= kProbeCullLights) {
                float pmaxD = maxabs3(mo);
                float pb1 = 0.f, pb2 = 0.f;
                for (int li = 0; li < S.n_lights; ++li) {
                  int lprim = __float_as_int(uniform_load4(S.lights + (size_t)li * kLightF4 + 4).w);
                  const float4* rec = S.isect + (size_t)lprim * kIsectF4;
                  float t, b1, b2;
                  bool last;
                  if (prim_test_rec(S, uniform_load4(rec), uniform_load4(rec + 1), uniform_load4(rec + 2), pos, mo, pmaxD,
                                    S.epsilon, INFINITY, t, b1, b2, last) &&
                      (t < ptmax || (t == ptmax && lprim > pseed) || pseed < 0)) {
                    ptmax = t;
                    pseed = lprim;
                    pb1 = b1; pb2 = b2;
                  }
                }
                if (pseed < 0) { want_probe = false; py = make_float4(0, 0, 0, 0); }
                else if (PROBE_ANY) {
                  // The emitter is known, so is the term it gives if nothing hides it: the probe
                  // becomes an occlusion ray whose hit clears the term (k_trace<true>), and a
                  // term that is zero anyway (an emitter seen from behind) needs no ray at all.
                  V3 y = probe_emitter_term(S, pseed, pos, mo, ptmax, pb1, pb2, mfr, mpdf, absdot);
                  py = f4(y, 0.f);
                  want_probe = !is_black(y);
                }
              }
            }
#endif
          }
          // Li += beta * (x + y) is owed once visibility and the probe are known.
          // Without a probe y counts as zero and is not stored; when x is zero
          // as well the sum adds beta * 0, which only matters for a non-finite
          // beta (the sample then turns NaN and is dropped), so a finite beta
          // stores nothing at all.
          bool finite_beta = fabsf(beta.x) < INFINITY && fabsf(beta.y) < INFINITY && fabsf(beta.z) < INFINITY;
          if (want_probe || !is_black(x) || !finite_beta) {
            P.pend_x[slot] = f4(x, absdot);
            if (want_probe) P.pend_y[slot] = py;
            P.beta_prev[slot] = f4(beta, 0.f);
            flags |= kFPending | (want_probe ? kFPendingY : 0u);
          }
        }
        // continuation
        if (is_black(fr2)) {
          alive = false;
        } else {
          beta = beta * (fr2 * fabsf(dot(nor, wo2)) / pdf2);
          specular = delta;
          no = pos; nd = wo2;
          if (bounces > 3) {
            float illumate = fps::clamp_(1.f - luminance(beta), 0.f, 1.f);
            if (rng_uniform(rng) < illumate) alive = false;
            else beta = beta / (1 - illumate);
          }
          if (bounces + 1 >= S.maxDepth) alive = false;
        }
      }
    }
    want_next = alive;
    flags = (flags & ~(kFBounceMask | kFSpecular)) | (uint32_t)(bounces + 1) | (specular ? kFSpecular : 0u);
    P.beta_rng[slot] = f4(beta, rng);
    P.li_flags[slot] = f4(Li, flags);
  }
  stage_push(stage.next, sc.base, sc.n_next, want_next, slot, no, -1, nd, INFINITY);
  stage_push(stage.shadow, sc.base_shadow, sc.n_shadow, want_shadow, slot, so, -1, sd, stmax);
  // Where every probe ray sets out knowing the nearest emitter on its line (few emitters, no
  // environment), whether that emitter is the closest hit is a question of occlusion: is there
  // a primitive in front of it, or at the same distance with a higher index?  Such a probe
  // joins the shadow rays (any hit, early exit, two launches less per bounce), marked by the
  // emitter's index where a shadow ray carries -1.
  if (PROBE_ANY) {
    stage_push(stage.shadow, sc.base_shadow, sc.n_shadow, want_probe, slot, po, pseed, pd, ptmax);
    sc.n_probe += (uint32_t)__popcll(lane_ballot(want_probe));
  } else {
    stage_push(stage.probe, sc.base, sc.n_probe, want_probe, slot, po, pseed, pd, ptmax);
  }
}

// The shade stage of the path tracer.
//
// Which queue entries need a vertex: a ray that hit nothing ends its path, and unless the scene has
// an environment light nothing is left to do for it (what the path still owes from its previous
// vertex is added by k_finish); in an open scene that is half the queue, and a vertex is some
// 3000 vector instructions that those lanes would sit through.  And what a vertex costs is mostly
// arithmetic of ITS material.  So a wavefront first sorts the entries of a round that need a
// vertex into short lists in LDS, one per material type (and one for the misses an environment
// light makes count), and shades a list whenever it holds 64: every lane of a group then has a
// vertex, and all 64 run the same material's code.  What is left in the lists when the
// wavefront's work is over is shaded as partial groups.  Measured (Msamples/s, one list for all
// hits against a list per material): staircase 897 / 912, torus knot 1448 / 1468, benchmark scene
// 1733 / 1734 -- most of the gain is in leaving out the misses, neighbouring paths mostly meet
// the same material anyway.
constexpr int kShadeLists = CASTER_SHADE_SORT ? 7 : 1;  // material types 0..5, misses under an environment light
constexpr int kShadeListCap = 128;                      // a list is emptied as soon as it reaches 64: 63 + 64 at most
template <bool PROBE_ANY>
__global__ __launch_bounds__(kWave, CASTER_SHADE_WAVES) void k_shade_pt(LaunchCtx L, RayQueue cur, ShadeQueues out, StageArea stage) {
  __shared__ uint32_t s_list[kShadeLists][kShadeListCap];
  const DevScene& S = L.S;
  const uint32_t count = *cur.count;
  const uint32_t lane = threadIdx.x;
  StageCursor sc{blockIdx.x * stage.region_cap, blockIdx.x * stage.shadow_region_cap, 0u, 0u, 0u};
  // ring buffers: lane c keeps the start and the length of list c
  uint32_t q_head = 0, q_len = 0;
  const bool env = S.inf.valid != 0;
  auto take_group = [&](int c, uint32_t n) {   // the first n (<= 64) entries of list c, one per lane
    uint32_t head = (uint32_t)__builtin_amdgcn_readlane((int)q_head, c);
    uint32_t my = s_list[c][(head + lane) & (kShadeListCap - 1)];
    if ((int)lane == c) { q_head = (q_head + n) & (kShadeListCap - 1); q_len -= n; }
    return my;
  };
  {
    CASTER_FOR_ROUNDS(r, count) {
      const uint32_t idx = r * kWave + lane;
      int cls = -1;
      if (idx < count) {
        int prim = __float_as_int(cur.hit[idx].y);
        if (prim >= 0) cls = CASTER_SHADE_SORT ? __float_as_int(S.shade[(size_t)prim * kShadeF4 + 7].y) : 0;
        else if (env && cur.d[idx].w >= 0.f) cls = CASTER_SHADE_SORT ? 6 : 0;   // (tmax < 0: a slot outside the image)
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
      unsigned long long full = lane_ballot(lane < (uint32_t)kShadeLists && q_len >= (uint32_t)kWave);
      while (full) {
        int c = __builtin_ctzll(full);
        uint32_t my = take_group(c, kWave);
        shade_pt_vertices<PROBE_ANY>(L, cur, stage, sc, my, true);
        full = lane_ballot(lane < (uint32_t)kShadeLists && q_len >= (uint32_t)kWave);
      }
      __syncthreads();
    }
  }
  // what is left in the lists
  for (int c = 0; c < kShadeLists; ++c) {
    uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)q_len, c);
    if (n == 0) continue;
    uint32_t my = take_group(c, n);
    shade_pt_vertices<PROBE_ANY>(L, cur, stage, sc, my, lane < n);
  }
  __syncthreads();  // the copy below reads what other lanes of this wavefront staged
  stage_flush(stage.next, sc.base, sc.n_next, out.next);
  stage_flush(stage.shadow, sc.base_shadow, sc.n_shadow, out.shadow);
  if (PROBE_ANY) {
    // the probe queue stays empty; its length counter keeps the number of probe rays for the statistics
    uint32_t n_probe = (uint32_t)__builtin_amdgcn_readfirstlane((int)sc.n_probe);
    if (lane == 0 && n_probe) atomicAdd(out.probe.count, n_probe);
  } else {
    stage_flush(stage.probe, sc.base, sc.n_probe, out.probe);
  }
}

// BSDF-sampled direct light (pathtracer.cu:949-983), after the probe rays have
// been traced: an emitter hit (or the environment on a miss) contributes
// weight * fr * Le * |cos| / pdf into the path's pending term.
__device__ inline void probe_resolve_one(const LaunchCtx& L, V3 pos, V3 out, Hit h, int slot) {
  const DevScene& S = L.S;
  bool found = h.prim >= 0;
  float4 py = L.P.pend_y[slot];
  V3 fr = v3(py);
  float pdf = py.w;
  float absdot = L.P.pend_x[slot].w;
  V3 y = v3(0, 0, 0);
  if (found) {
    // a hit the ray started out with carries no barycentrics: the same test gives them again
    float t, b1, b2;
    bool last;
    if (prim_test(S, h.prim, pos, out, maxabs3(out), S.epsilon, INFINITY, t, b1, b2, last)) {
      h.b1 = b1; h.b2 = b2;
    }
    y = probe_emitter_term(S, h.prim, pos, out, h.t, h.b1, h.b2, fr, pdf, absdot);
  } else if (S.inf.valid) {
    V3 radiance = inf_le(S.inf, out);
    float choicePdf = S.light_cdf[S.n_lights + 1] - S.light_cdf[S.n_lights];
    float lightPdf = ONE_OVER_FOUR_PI;
    float weight = power_heuristic(pdf, lightPdf * choicePdf);
    y = weight * fr * radiance * absdot / pdf;
  }
  L.P.pend_y[slot] = f4(y, 0.f);
}
__global__ __launch_bounds__(kWave) void k_probe_resolve(LaunchCtx L, RayQueue q) {
  const uint32_t count = *q.count;
  for (uint32_t idx = blockIdx.x * kWave + threadIdx.x; idx < count; idx += gridDim.x * kWave) {
    float4 o4 = q.o[idx], d4 = q.d[idx], h4 = q.hit[idx];
    Hit h;
    h.t = h4.x; h.prim = __float_as_int(h4.y); h.b1 = h4.z; h.b2 = h4.w;
    probe_resolve_one(L, v3(o4), v3(d4), h, q.slot[idx]);
  }
}
// A vertex whose probe ray was never issued does not set kFPendingY, so its
// pend_y is not read; issued ones are always overwritten by k_probe_resolve.

// One vertex of Ao() (pathtracer.cu:841-868).
__global__ __launch_bounds__(kWave) void k_shade_ao(LaunchCtx L, RayQueue cur, ShadeQueues out, StageArea stage) {
  const DevScene& S = L.S;
  const PathState& P = L.P;
  const uint32_t count = *cur.count;
  const uint32_t sbase = blockIdx.x * stage.region_cap;
  uint32_t n_next = 0, n_shadow = 0, n_probe = 0;
  CASTER_FOR_ROUNDS_AO(r, count) {
    uint32_t idx = r * kWave + threadIdx.x;
    bool want_shadow = false;
    int slot = 0;
    V3 so = v3(0, 0, 0), sd = v3(0, 0, 1);
    if (idx < count) {
      slot = cur.slot[idx];
      float4 o4 = cur.o[idx], d4 = cur.d[idx], h4 = cur.hit[idx];
      if (d4.w >= 0.f) {
        V3 ro = v3(o4), rd = v3(d4);
        Hit h;
        h.t = h4.x; h.prim = __float_as_int(h4.y); h.b1 = h4.z; h.b2 = h4.w;
        float4 br = P.beta_rng[slot];
        uint32_t rng = __float_as_uint(br.w);
        uint32_t flags = 1u;
        V3 Li = v3(0, 0, 0);
        if (h.prim < 0) {
          Li = v3(1, 1, 1);
        } else {
          Isect is;
          load_isect(S, ro, rd, h, is);
          V3 n = is.n;
          float pdf = 0.f;
          if (dot(-rd, n) < 0.f) n = -n;
          float u1 = rng_uniform(rng);
          float u2 = rng_uniform(rng);
          V3 wi = cosine_hemisphere(u1, u2, pdf);
          V3 s = is.dpdu;
          V3 t = cross(n, s);
          wi = to_world(wi, s, t, n);
          float cosine = dot(wi, n);
          float v = cosine / pdf;
          P.pend_x[slot] = f4(v3(v, v, v), 0.f);
          P.beta_prev[slot] = f4(v3(1, 1, 1), 0.f);
          flags |= kFPending;
          want_shadow = true;
          so = is.pos; sd = wi;
        }
        P.beta_rng[slot] = f4(v3(br), rng);
        P.li_flags[slot] = f4(Li, flags);
      }
    }
    stage_push(stage.shadow, sbase, n_shadow, want_shadow, slot, so, -1, sd, S.maxDist);
  }
  __syncthreads();
  stage_flush(stage.shadow, sbase, n_shadow, out.shadow);
  (void)n_next; (void)n_probe;
}

}  // namespace caster
