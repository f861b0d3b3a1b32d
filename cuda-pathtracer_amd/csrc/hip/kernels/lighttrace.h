// Part of the gfx950 wavefront renderer (included by ../kernels.hip, in this order: common.h, traverse.h, trace.h, shading.h, bsdf.h, path.h, volpath.h, lighttrace.h, bdpt.h, volpath_queues.h, film.h).
// Light tracing (k_generate_lt, k_shade_lt, k_lt_resolve).
#pragma once

namespace caster {

// ---- light tracing ------------------------------------------------------------
// LightTracing() (pathtracer.cu:1240-1375): paths start on the emitters and every vertex is
// connected to the pinhole and splatted onto the film.  The connections' transmittance walks run
// inside these kernels, lane by lane, as in k_shade_vpt; only the continuation ray goes back
// through the extend stage.  Splats are float atomics straight into the film sum, so the order
// in which a pixel's splats are added, and with it the last bits of the image, varies from run
// to run (as it does in the reference).

// Camera::SampleCamera (Camera.h:87-118).  pdf == 0: the pinhole does not see `pos`.
__device__ inline void sample_camera(const DevScene& S, V3 pos, V3& sdir, float& stmax, float& we, float& pdf,
                                     int& raster) {
  const DevCamera& c = S.cam;
  V3 dir = v3(c.pos) - pos;
  V3 ndir = normalize(dir);
  sdir = ndir;
  stmax = length(dir) - S.epsilon;
  V3 negative_dir = -ndir;
  V3 cndir = v3(dot(negative_dir, v3(c.u)), dot(negative_dir, v3(c.v)), dot(negative_dir, v3(c.w)));
  pdf = 0.f;
  if (cndir.z >= 0.f) return;
  float costheta = dot(cndir, v3(0, 0, -1.f));
  float scale = -c.distance / cndir.z;
  cndir = cndir * scale;
  V2 plane = V2{cndir.x / c.half_w, cndir.y / c.half_h};
  if (plane.x > 1.f || plane.x < -1.f || plane.y > 1.f || plane.y < -1.f) return;
  plane = V2{plane.x * 0.5f + 0.5f, plane.y * 0.5f + 0.5f};
  int x = fps::f2i_(floorf(plane.x * (c.res_x - 1) + 0.5f));
  int y = fps::f2i_(floorf(plane.y * (c.res_y - 1) + 0.5f));
  raster = fps::f2i_(y * c.res_x + x);
  pdf = dot(dir, dir) / costheta;
  we = c.distance * c.distance / (c.area * costheta * costheta * costheta * costheta);
}
__device__ inline void splat(const DevScene& S, float* acc, int raster, V3 v) {
  if (raster < 0 || raster >= S.width * S.height) return;  // a NaN position
  // adding a zero changes nothing (the sums start at +0), and a third to a half of all connections
  // are occluded or face away from the camera: their atomics are not issued
  if (v.x != 0.f) unsafeAtomicAdd(acc + 3 * (size_t)raster, v.x);
  if (v.y != 0.f) unsafeAtomicAdd(acc + 3 * (size_t)raster + 1, v.y);
  if (v.z != 0.f) unsafeAtomicAdd(acc + 3 * (size_t)raster + 2, v.z);
}
// Medium::Sample (medium.h:18-38 homogeneous, :81-109 delta tracking): scales beta, reports a
// scattering event before the surface at rtmax.
__device__ inline void medium_sample(const DevScene& S, int medium, const MediumRec& med, V3 ro, V3 rd, float rtmax,
                                     uint32_t& rng, V3& beta, float& sampledDist, bool& sampledMedium) {
  if (med.type == 0) {
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
    return;
  }
  GridRec g;
  load_grid(S, medium, g);
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
      return;
    }
    if (--iter == 0) break;
  }
  beta = beta * v3(1.f, 1.f, 1.f);
}
// Medium::Phase and Medium::SamplePhase (medium.h:146-186)
__device__ inline float phase_eval(const MediumRec& med, V3 in, V3 out) {
  if (med.g == 0) return ONE_OVER_FOUR_PI;
  float costheta = dot(in, out);
  float cubicTerm = (1.f + med.g * med.g - 2.f * med.g * costheta);
  return ONE_OVER_FOUR_PI * (1.f - med.g * med.g) / sqrtf(cubicTerm * cubicTerm * cubicTerm);
}
__device__ inline V3 phase_sample(const MediumRec& med, float ux, float uy) {
  if (med.g == 0) {
    float pdf;
    return uniform_sphere(ux, uy, pdf);
  }
  float g = med.g, costheta;
  if (fabsf(g) < 1e-3f) costheta = 1.f - 2.f * ux;
  else {
    float sqrtTerm = (1.f - g * g) / (1.f - g + 2.f * g * ux);
    costheta = (1.f + g * g - sqrtTerm * sqrtTerm) / (2.f * g);
  }
  float sintheta = sqrtf(1.f - costheta * costheta);
  float phi = TWOPI * uy;
  float sinphi = fps::sin_(phi), cosphi = fps::cos_(phi);
  return v3(sintheta * cosphi, sintheta * sinphi, costheta);
}

// Start of every light path of iterations [first_iter, first_iter + n_iter) (pathtracer.cu:
// 1245-1272): seed, light choice, a point and a cosine-distributed direction on the chosen
// triangle (area.h:19-24, Triangle.h:157-166), and the connection of that point to the camera.
// A path that picks no emissive triangle (none in the scene, or the environment's bin, where the
// reference reads past its light array) draws its numbers and carries nothing.  `acc` null: no
// connection, only the rays (for the first-hit map).
// QUEUED (scenes without a density grid, where Tr() draws no random numbers): a connection is not
// walked here but sent through the traversal kernel -- its ray goes to `conn.shadow`, the path's
// slot to the list `conn.probe`, the operands of its contribution to pend_x / pend_y / pend_z, and
// k_lt_resolve splats it once k_tr_step / k_tr_finish have left its transmittance in pend_w.
struct LtConn {
  RayQueue shadow, probe;  // final queues
  StageArea stage;
};
enum { kLtSurface = 0, kLtMedium = 1, kLtStart = 2 };
template <bool QUEUED>
__global__ __launch_bounds__(kWave) void k_generate_lt(LaunchCtx L, RayQueue q, uint32_t first_iter, uint32_t n_iter,
                                                       float* acc, WalkAux aux, LtConn conn) {
  __shared__ uint2 s_stack[kLdsDepth][kWave];
  Stack st{s_stack, aux.spill + (blockIdx.x * kWave + threadIdx.x), aux.spill_stride, 0};
  const DevScene& S = L.S;
  uint32_t stat_tr = 0;
  uint32_t total = (uint32_t)S.slots_per_iter * n_iter;  // a multiple of the wavefront size
  const uint32_t sbase = blockIdx.x * conn.stage.region_cap;
  uint32_t n_conn = 0, n_list = 0;
  for (uint32_t idx = blockIdx.x * kWave + threadIdx.x; idx < total; idx += gridDim.x * kWave) {
    bool want_conn = false;
    V3 co = v3(0, 0, 0), cd = v3(0, 0, 1);
    float ctm = 0.f;
    uint32_t k, local;
    batch_slot_split(S, idx, n_iter, k, local);
    uint32_t px, py;
    bool live = slot_pixel(S, local, px, py);
    uint32_t iter = first_iter + k;
    float4 o4 = make_float4(0, 0, 0, __int_as_float(-1)), d4 = make_float4(0, 0, 1, -1.f);
    uint32_t rng = 1u, flags = 0;
    V3 beta = v3(1.f, 1.f, 1.f);
    if (live) {
      uint32_t pixel = px + py * (uint32_t)S.width;
      rng = rng_seed(pixel, iter);
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
        beta = beta * (l.radiance * fabsf(dot(dir, nor)) / (pdfA * pdfW * choicePdf));
        if (acc) {
          V3 cdir;
          float ctmax, we = 0.f, cameraPdf = 0.f;
          int raster = 0;
          sample_camera(S, pos, cdir, ctmax, we, cameraPdf, raster);
          if (cameraPdf != 0.f) {
            if (QUEUED) {
              want_conn = true;
              co = pos; cd = cdir; ctm = ctmax;
              L.P.pend_x[idx] = f4(l.radiance, 0.f);
              L.P.pend_z[idx] = make_float4(cameraPdf, __int_as_float(raster), __int_as_float(kLtStart), 0.f);
              L.P.pend_w[idx] = f4(v3(1.f, 1.f, 1.f), (uint32_t)lmed);
            } else {
              V3 tr = transmittance(S, pos, cdir, S.epsilon, ctmax, lmed, st, L.stats, stat_tr, rng);
              if (!is_black(tr)) splat(S, acc, raster, tr * l.radiance);
            }
          }
        }
        o4 = f4(pos, __int_as_float(-1));
        d4 = f4(dir, INFINITY);
        flags = (uint32_t)(lmed + 1) << kFMediumShift;
      }
    }
    q.o[idx] = o4;
    q.d[idx] = d4;
    q.slot[idx] = (int)idx;
    L.P.beta_rng[idx] = f4(beta, rng);
    L.P.li_flags[idx] = f4(v3(0.f, 0.f, 0.f), flags);
    L.P.beta_prev[idx] = make_float4(0, 0, 0, 0);
    if (QUEUED) {
      stage_push(conn.stage.shadow, sbase, n_conn, want_conn, (int)idx, co, -1, cd, ctm);
      stage_push(conn.stage.probe, sbase, n_list, want_conn, (int)idx, co, -1, cd, ctm);
    }
  }
  if (QUEUED) {
    __syncthreads();
    stage_flush(conn.stage.shadow, sbase, n_conn, conn.shadow);
    stage_flush(conn.stage.probe, sbase, n_list, conn.probe);
  }
  for (int off = 32; off > 0; off >>= 1) stat_tr += (uint32_t)__shfl_down((int)stat_tr, off, 64);
  if ((threadIdx.x & 63u) == 0 && stat_tr) atomicAdd(&L.stats->rays[3], (unsigned long long)stat_tr);
}

// One event of a light path (pathtracer.cu:1275-1374).
// (the kernel that walks its connections in place takes 149 registers, three wavefronts per SIMD; held to the budget of
// four it spills 32 values and the light tracer through a density grid runs 194 -> 217 Msamples/s, the Cornell box the
// same; the queued kernel fits four as it is)
#ifndef CASTER_LT_WAVES
#define CASTER_LT_WAVES 4
#endif
template <bool QUEUED>
__global__ __launch_bounds__(kWave, QUEUED ? 1 : CASTER_LT_WAVES) void k_shade_lt(LaunchCtx L, RayQueue cur, ShadeQueues out, StageArea stage,
                                                    WalkAux aux, float* acc) {
  __shared__ uint2 s_stack[kLdsDepth][kWave];
  Stack st{s_stack, aux.spill + (blockIdx.x * kWave + threadIdx.x), aux.spill_stride, 0};
  const DevScene& S = L.S;
  const PathState& P = L.P;
  const uint32_t count = *cur.count;
  uint32_t stat_tr = 0;
  const uint32_t sbase = blockIdx.x * stage.region_cap;
  uint32_t n_next = 0, n_conn = 0, n_list = 0;
  CASTER_FOR_ROUNDS_LT(r, count) {
    uint32_t idx = r * kWave + threadIdx.x;
    bool want_next = false, want_conn = false;
    int slot = 0;
    V3 no = v3(0, 0, 0), nd = v3(0, 0, 1), co = no, cd = nd;
    float ctm = 0.f;
    if (idx < count) {
      slot = cur.slot[idx];
      float4 o4 = cur.o[idx], d4 = cur.d[idx], h4 = cur.hit[idx];
      if (d4.w >= 0.f || __float_as_int(h4.y) >= 0) {
        V3 ro = v3(o4), rd = v3(d4);
        Hit h;
        h.t = h4.x; h.prim = __float_as_int(h4.y); h.b1 = h4.z; h.b2 = h4.w;
        float4 br = P.beta_rng[slot], lf = P.li_flags[slot];
        V3 beta = v3(br);
        uint32_t rng = __float_as_uint(br.w), flags = __float_as_uint(lf.w);
        int bounces = (int)(flags & kFBounceMask);
        int rmed = (int)((flags & kFMediumMask) >> kFMediumShift) - 1;
        int crossings = (int)__float_as_uint(P.beta_prev[slot].w);
        int nmed = rmed;
        bool alive = true;
        bool count_bounce = true;
        if (h.prim < 0) {
          alive = false;
        } else {
          Isect is;
          load_isect(S, ro, rd, h, is);
          V3 pos = is.pos, nor = is.n, dpdu = is.dpdu;
          V2 uv = is.uv;
          float rtmax = h.t;
          float sampledDist = 0.f;
          bool sampledMedium = false;
          MediumRec med;
          if (rmed != -1) {
            load_medium(S, rmed, med);
            medium_sample(S, rmed, med, ro, rd, rtmax, rng, beta, sampledDist, sampledMedium);
          }
          if (is_black(beta)) {
            alive = false;
          } else if (sampledMedium) {
            V3 samplePos = ro + sampledDist * rd;
            V3 cdir;
            float ctmax, we = 0.f, cameraPdf = 0.f;
            int raster = 0;
            sample_camera(S, samplePos, cdir, ctmax, we, cameraPdf, raster);
            // the reference walks Tr() whatever the camera sees (pathtracer.cu:1299); the walk draws
            // random numbers only inside a density grid, so without one a connection that cannot
            // contribute skips it
            if (QUEUED) {
              if (cameraPdf != 0.f) {
                want_conn = true;
                co = samplePos; cd = cdir; ctm = ctmax;
                P.pend_x[slot] = f4(beta, we);
                P.pend_y[slot] = make_float4(phase_eval(med, -rd, cdir), 0.f, 0.f, 0.f);
                P.pend_z[slot] = make_float4(cameraPdf, __int_as_float(raster), __int_as_float(kLtMedium), 0.f);
                P.pend_w[slot] = f4(v3(1.f, 1.f, 1.f), (uint32_t)rmed);
              }
            } else if (cameraPdf != 0.f || S.has_hetero) {
              V3 tr = transmittance(S, samplePos, cdir, S.epsilon, ctmax, rmed, st, L.stats, stat_tr, rng);
              float phase = phase_eval(med, -rd, cdir);
              if (cameraPdf != 0.f) {
                V3 Lc = beta * we * tr * phase / cameraPdf;
                if (!has_inf(Lc) && !has_nan(Lc)) splat(S, acc, raster, Lc);
              }
            }
            float ux = rng_uniform(rng);
            float uy = rng_uniform(rng);
            no = samplePos; nd = phase_sample(med, ux, uy); nmed = rmed;
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
              V3 cdir;
              float ctmax, we = 0.f, cameraPdf = 0.f;
              int raster = 0;
              sample_camera(S, pos, cdir, ctmax, we, cameraPdf, raster);
              if (cameraPdf != 0.f) {
                V3 fr;
                float unuse;
                if (QUEUED) {
                  eval_bsdf(S, m, -rd, cdir, nor, uv, dpdu, fr, unuse);
                  want_conn = true;
                  co = pos; cd = cdir; ctm = ctmax;
                  P.pend_x[slot] = f4(beta, we);
                  P.pend_y[slot] = f4(fr, fabsf(dot(cdir, nor)));
                  P.pend_z[slot] = make_float4(cameraPdf, __int_as_float(raster), __int_as_float(kLtSurface), 0.f);
                  P.pend_w[slot] = f4(v3(1.f, 1.f, 1.f), (uint32_t)rmed);
                } else {
                  V3 tr = transmittance(S, pos, cdir, S.epsilon, ctmax, rmed, st, L.stats, stat_tr, rng);
                  eval_bsdf(S, m, -rd, cdir, nor, uv, dpdu, fr, unuse);
                  V3 Lc = tr * beta * fr * we * fabsf(dot(cdir, nor)) / cameraPdf;
                  if (!has_inf(Lc) && !has_nan(Lc)) splat(S, acc, raster, Lc);
                }
              }
            }
            V3 u;
            u.x = rng_uniform(rng); u.y = rng_uniform(rng); u.z = rng_uniform(rng);
            V3 wo2, fr2;
            float pdf2;
            sample_bsdf(S, m, -rd, nor, uv, dpdu, u, wo2, fr2, pdf2, false);
            if (is_black(fr2)) {
              alive = false;
            } else {
              beta = beta * (fr2 * fabsf(dot(wo2, nor)) / pdf2);
              int mm = dot(wo2, nor) > 0 ? is.med_out : is.med_in;
              mm = dot(-rd, nor) * dot(wo2, nor) > 0 ? rmed : mm;
              nmed = mm;
              no = pos; nd = wo2;
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
        flags = (uint32_t)bounces | ((uint32_t)(nmed + 1) << kFMediumShift);
        P.beta_rng[slot] = f4(beta, rng);
        P.li_flags[slot] = f4(v3(0.f, 0.f, 0.f), flags);
        P.beta_prev[slot] = make_float4(0, 0, 0, __uint_as_float((uint32_t)crossings));
      }
    }
    stage_push(stage.next, sbase, n_next, want_next, slot, no, -1, nd, INFINITY);
    if (QUEUED) {
      stage_push(stage.shadow, sbase, n_conn, want_conn, slot, co, -1, cd, ctm);
      stage_push(stage.probe, sbase, n_list, want_conn, slot, co, -1, cd, ctm);
    }
  }
  __syncthreads();
  stage_flush(stage.next, sbase, n_next, out.next);
  if (QUEUED) {
    stage_flush(stage.shadow, sbase, n_conn, out.shadow);
    stage_flush(stage.probe, sbase, n_list, out.probe);
  }
  for (int off = 32; off > 0; off >>= 1) stat_tr += (uint32_t)__shfl_down((int)stat_tr, off, 64);
  if ((threadIdx.x & 63u) == 0 && stat_tr) atomicAdd(&L.stats->rays[3], (unsigned long long)stat_tr);
}

// The connections of one round of light tracing on queues, their transmittance known: the same
// products, in the same order, as the in-kernel path above forms (pathtracer.cu:1271, 1303, 1343).
__global__ __launch_bounds__(kWave) void k_lt_resolve(LaunchCtx L, RayQueue list, float* acc) {
  const DevScene& S = L.S;
  const PathState& P = L.P;
  const uint32_t count = *list.count;
  for (uint32_t idx = blockIdx.x * kWave + threadIdx.x; idx < count; idx += gridDim.x * kWave) {
    int slot = list.slot[idx];
    float4 a = P.pend_x[slot], c = P.pend_z[slot];
    V3 tr = v3(P.pend_w[slot]);
    int raster = __float_as_int(c.y), kind = __float_as_int(c.z);
    float cameraPdf = c.x;
    if (kind == kLtStart) {
      if (!is_black(tr)) splat(S, acc, raster, tr * v3(a));
      continue;
    }
    float4 b = P.pend_y[slot];
    V3 beta = v3(a);
    float we = a.w;
    V3 Lc = kind == kLtMedium ? beta * we * tr * b.x / cameraPdf : tr * beta * v3(b) * we * b.w / cameraPdf;
    if (!has_inf(Lc) && !has_nan(Lc)) splat(S, acc, raster, Lc);
  }
}

}  // namespace caster
