// Part of the gfx950 wavefront renderer (included by ../kernels.hip, in this order: common.h, traverse.h, trace.h, shading.h, bsdf.h, path.h, volpath.h, lighttrace.h, bdpt.h, volpath_queues.h, film.h).
// Launch context, batch layout, camera samples (k_generate), the persistent traversal kernel (k_trace).
#pragma once

namespace caster {

// ----------------------------------------------------------------- kernels
struct LaunchCtx {
  DevScene S;
  PathState P;
  Stats* stats;
};

// Slot of (iteration k of the batch, pixel slot `local` of one iteration).  The
// batch is laid out tile by tile: the n_iter wavefronts (8x8 pixels of one
// iteration each) of a tile follow one another, so that a contiguous part of a
// queue -- what one XCD's wavefronts consume -- covers a band of the image
// rather than a few iterations of all of it, and that XCD's L2 sees the
// geometry of that band.
#ifndef CASTER_TILE_MAJOR
#define CASTER_TILE_MAJOR 1
#endif
__device__ inline uint32_t batch_slot(const DevScene& S, uint32_t k, uint32_t local, uint32_t n_iter) {
#if CASTER_TILE_MAJOR
  return (((local >> 6) * n_iter + k) << 6) | (local & 63u);
#else
  return k * (uint32_t)S.slots_per_iter + local;
#endif
}
__device__ inline void batch_slot_split(const DevScene& S, uint32_t slot, uint32_t n_iter, uint32_t& k, uint32_t& local) {
#if CASTER_TILE_MAJOR
  uint32_t w = slot >> 6, tile = w / n_iter;
  k = w - tile * n_iter;
  local = (tile << 6) | (slot & 63u);
#else
  k = slot / (uint32_t)S.slots_per_iter;
  local = slot - k * (uint32_t)S.slots_per_iter;
#endif
}

// Camera samples for iterations [first_iter, first_iter + n_iter): seeds the
// RNG, draws the four camera numbers (pathtracer.cu:881-890), writes the
// extension ray and fresh path state.  Queue index == path slot here.
__global__ __launch_bounds__(kWave) void k_generate(LaunchCtx L, RayQueue q, uint32_t first_iter, uint32_t n_iter,
                                                    int y0, int y1) {
  const DevScene& S = L.S;
  uint32_t total = (uint32_t)S.slots_per_iter * n_iter;
  for (uint32_t idx = blockIdx.x * kWave + threadIdx.x; idx < total; idx += gridDim.x * kWave) {
    uint32_t k, local;
    batch_slot_split(S, idx, n_iter, k, local);
    uint32_t px, py;
    bool live = slot_pixel(S, local, px, py) && (int)py >= y0 && (int)py < y1;
    uint32_t iter = first_iter + k;
    float4 o4 = make_float4(0, 0, 0, __int_as_float(-1)), d4 = make_float4(0, 0, 1, -1.f);
    uint32_t rng = 1u;
    if (live) {
      uint32_t pixel = px + py * (uint32_t)S.width;
      rng = rng_seed(pixel, iter);
      float offsetx = rng_uniform(rng) - 0.5f;
      float offsety = rng_uniform(rng) - 0.5f;
      // Bdpt() draws no lens sample and shoots through the centre of the aperture (pathtracer.cu:1403-1408)
      float a1 = S.integrator == 4 /*IT_BDPT*/ ? 0.f : rng_uniform(rng);
      float a2 = S.integrator == 4 ? 0.f : rng_uniform(rng);
      V3 orig, dir;
      primary_ray(S.cam, px + offsetx, py + offsety, a1, a2, orig, dir);
      o4 = f4(orig, __int_as_float(-1));
      d4 = f4(dir, INFINITY);
    }
    uint32_t flags = 0;
    if (S.integrator == 2 /*IT_VPT*/) flags |= (uint32_t)(S.cam.medium + 1) << kFMediumShift;
    q.o[idx] = o4;
    q.d[idx] = d4;  // tmax < 0 marks a slot outside the image: extend skips it
    q.slot[idx] = (int)idx;
    L.P.beta_rng[idx] = f4(v3(1.f, 1.f, 1.f), rng);
    L.P.li_flags[idx] = f4(v3(0.f, 0.f, 0.f), flags);
    L.P.beta_prev[idx] = make_float4(0, 0, 0, 0);
  }
}

__global__ void k_set_count(uint32_t* c, uint32_t v) { *c = v; }

// Traces every ray of a queue with persistent wavefronts.
//
// A wavefront claims rays a chunk at a time from the queue's work counter and
// keeps its 64 lanes fed: when kRefillIdle lanes have finished their rays the
// idle lanes take the next unclaimed ones, so a long ray does not hold 63 idle
// lanes hostage.  Between refills every step does the kind of work (a 4-wide
// node, or one primitive test) that more lanes are waiting for.
//   ANY=false: closest hit; (t, prim, b1, b2) goes into the queue entry.
//   ANY=true:  any hit; an occluded ray clears the pending light-sampled term
//              of its path.
#ifndef CASTER_SPECULATE
#define CASTER_SPECULATE 1
#endif
#ifndef CASTER_ANY_UNSORTED
#define CASTER_ANY_UNSORTED 1
#endif
#ifndef CASTER_TRACE_WAVES
#define CASTER_TRACE_WAVES 5
#endif
#ifdef CASTER_WALK_STATS
// tuning aid: wavefront-level step counts of k_trace, [ANY][counter]
__device__ unsigned long long g_walk_stats[2][8];
#define CASTER_WS(i, v) ws[i] += (unsigned long long)(v)
#else
#define CASTER_WS(i, v)
#endif
template <bool ANY>
__global__ __launch_bounds__(kWave * kTraceBlockWaves, CASTER_TRACE_WAVES) void k_trace(LaunchCtx L, RayQueue q, WalkAux aux) {
  __shared__ uint2 s_stack[kTraceBlockWaves][kLdsDepth][kWave];
  __shared__ float4 s_top[kTopNodes > 0 ? kTopNodes * kTopF4 : 1];
#ifdef CASTER_WALK_STATS
  unsigned long long ws[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  const DevScene& S = L.S;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave_in_block = threadIdx.x >> 6;
  const uint32_t wave_id = blockIdx.x * kTraceBlockWaves + wave_in_block;  // index of this wavefront in the grid
  const uint32_t count = *q.count;
  unsigned long long* overflow = &L.stats->stack_overflow;
  // the top of the tree, shared by the workgroup's wavefronts
  const int n_top = S.n_nodes < kTopNodes ? S.n_nodes : kTopNodes;
  if (kTopNodes > 0) {
    for (int i = (int)threadIdx.x; i < n_top * kTopF4; i += kWave * kTraceBlockWaves)
      s_top[i] = S.nodes[(size_t)(i / kTopF4) * kNodeF4 + (i % kTopF4)];
    __syncthreads();
  }
  Stack st{s_stack[wave_in_block], aux.spill + (wave_id * kWave + lane), aux.spill_stride, 0};
  int idx = -1;
  int cur = kRefEmpty;
  int pend = kRefEmpty;  // a leaf waiting for its primitives to be tested
  V3 o = v3(0, 0, 0), d = v3(0, 0, 1), inv = v3(0, 0, 1);
  const float tmin = S.epsilon;
  float maxD = 1.f;
  Hit hit;
  hit.t = 0.f; hit.prim = -1; hit.b1 = hit.b2 = 0.f;
  bool occluded = false;
  // this wavefront's first chunk is assigned statically (no atomic storm at
  // launch): the work counters start at the end of the static assignment
  const uint32_t part = ((count + kWorkShards - 1) / kWorkShards + kWave - 1) / kWave * kWave;
  // Rays are claimed kChunk at a time from a short queue (so that they spread over all wavefronts)
  // and up to four times as many at a time from a long one (fewer trips to the work counters).
  const uint32_t n_waves = gridDim.x * kTraceBlockWaves;
  const uint32_t per_wave = count / (n_waves * kChunk);
  const uint32_t chunk = per_wave >= 16 ? 4 * kChunk : per_wave >= 8 ? 2 * kChunk : kChunk;
  // the work counters count from the end of the statically assigned first chunk of every wavefront
  const uint32_t work_init = (n_waves + kWorkShards - 1) / kWorkShards * chunk;
  uint32_t shard = wave_id % kWorkShards, shards_tried = 0;
  uint32_t chunk_next, chunk_end;
  {
    uint32_t lo = shard * part, hi = lo + part < count ? lo + part : count;
    uint32_t base = (wave_id / kWorkShards) * chunk;
    chunk_next = lo + base < hi ? lo + base : 0;
    chunk_end = lo + base < hi ? (chunk_next + chunk < hi ? chunk_next + chunk : hi) : 0;
  }
  bool more = true;
  for (;;) {
    // retire finished rays
    if (idx >= 0 && cur == kRefEmpty && pend == kRefEmpty) {
      if (ANY) {
        if (occluded) {
          int slot = q.slot[idx];
          if (hit.prim < 0) {
            float4 px = L.P.pend_x[slot];
            px.x = 0.f; px.y = 0.f; px.z = 0.f;
            L.P.pend_x[slot] = px;
          } else {
            // a probe ray among the shadow rays: something lies in front of the emitter it was
            // sent to, and the BSDF-sampled term the vertex stored for that emitter is void
            L.P.pend_y[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
      } else {
        q.hit[idx] = make_float4(hit.t, __int_as_float(hit.prim), hit.b1, hit.b2);
      }
      idx = -1;
    }
    // refill idle lanes
    unsigned long long idle_mask = __ballot(idx < 0);
    uint32_t n_idle = (uint32_t)__popcll(idle_mask);
    if (more && n_idle >= (uint32_t)kRefillIdle) {
      if (chunk_next == chunk_end) {
        // Claim the next chunk.  The queue is cut into kWorkShards contiguous
        // parts with a work counter each: same-address atomics with a return
        // value retire at only ~15 per microsecond on this chip, so one shared
        // counter would bound the kernel.  A wavefront works on "its" part
        // first and moves on to the next part when that one is exhausted.
        while (shards_tried < kWorkShards) {
          uint32_t lo = shard * part, hi = lo + part < count ? lo + part : count;
          // A part whose rays were all handed out statically needs no atomic to be found empty: in a
          // launch over a short queue that is every part, and 5120 wavefronts asking eight counters
          // in turn would take longer than the tracing (same-address atomics retire at ~20 per us).
          if (lo + work_init < hi) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(aux.work + shard * kWorkPad, chunk);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base) + work_init;
            if (lo + base < hi) {
              chunk_next = lo + base;
              chunk_end = chunk_next + chunk < hi ? chunk_next + chunk : hi;
              break;
            }
          }
          shard = shard + 1 == kWorkShards ? 0 : shard + 1;
          ++shards_tried;
        }
        if (shards_tried == kWorkShards) more = false;
      }
      if (more) {
        uint32_t avail = chunk_end - chunk_next;
        uint32_t give = avail < n_idle ? avail : n_idle;
        uint32_t prefix = (uint32_t)__popcll(idle_mask & ((1ull << lane) - 1ull));
        if (idx < 0 && prefix < give) {
          idx = (int)(chunk_next + prefix);
          float4 o4 = q.o[idx], d4 = q.d[idx];
          o = v3(o4); d = v3(d4);
          inv = v3(1.f / d.x, 1.f / d.y, 1.f / d.z);
          maxD = maxcomp(vabs(d));
          // every queued ray starts at the scene's epsilon; o.w carries a hit known
          // beforehand (a probe ray's nearest emitter, at distance tmax) or -1
          hit.t = d4.w; hit.prim = __float_as_int(o4.w); hit.b1 = hit.b2 = 0.f;
          occluded = false;
          st.sp = 0;
          // tmax < 0 marks a camera slot outside the image: nothing to trace.  (A tmax that is not
          // a number -- a light sample at the vertex itself -- bounds nothing: every comparison with
          // it fails, in the slab test and in the primitive tests, and the ray is traced unbounded,
          // as the reference traces it.)
          cur = (!(d4.w < 0.f) && S.n_nodes) ? 0 : kRefEmpty;
        }
        chunk_next += give;
      }
    }
    CASTER_WS(6, 1);
    if (!__any(idx >= 0)) {
      if (!more) break;
      continue;
    }
    // Advance until enough lanes are free again.  Each step the wavefront does
    // whichever kind of work more of its lanes can do: one 4-wide node for every
    // lane that stands at an inner node, or one primitive test for every lane
    // that holds a leaf.  A lane that reaches a leaf parks it in `pend` (a leaf
    // reference is ~index of the next primitive to test, so the cursor lives in
    // the reference itself) and goes on with the next deferred node, so that it
    // keeps taking part in node steps; the parked leaf may later shorten the ray
    // and make some of that node work unnecessary, never wrong.
    for (;;) {
#if CASTER_SPECULATE
      if (cur < 0 && cur != kRefEmpty && pend == kRefEmpty) {
        pend = cur;
        cur = stack_next(st, lane, hit.t);
      }
      unsigned long long m_inner = __ballot(cur >= 0);
      unsigned long long m_leaf = __ballot(pend != kRefEmpty);
      unsigned long long m_busy = __ballot(cur != kRefEmpty || pend != kRefEmpty);
      int n_inner = __popcll(m_inner), n_leaf = __popcll(m_leaf);
      int n_free = kWave - __popcll(m_busy);
      if (m_busy == 0) break;
      if (more && n_free >= kRefillIdle) break;
      CASTER_WS(4, __popcll(m_busy));
      CASTER_WS(5, __popcll(m_inner & m_leaf));
      if (n_inner * CASTER_VOTE_NUM >= n_leaf * CASTER_VOTE_DEN) {
        CASTER_WS(0, 1);
        CASTER_WS(1, n_inner);
        if (cur >= 0) cur = node_step<!ANY || !CASTER_ANY_UNSORTED>(S, cur, o, inv, hit.t, st, lane, overflow, s_top, n_top);
      } else if (pend != kRefEmpty) {
        CASTER_WS(2, 1);
        CASTER_WS(3, n_leaf);
        int i = ~pend;
        float t, b1, b2;
        bool last;
        bool found = prim_test(S, i, o, d, maxD, tmin, hit.t, t, b1, b2, last);
        if (found) {
          // (an occlusion ray holds -1 in hit.prim, or the emitter a probe ray was sent to, at
          // distance hit.t: the emitter itself and what it hides at equal distance do not occlude)
          if (ANY) {
            if (t < hit.t || i > hit.prim) occluded = true;
          } else if (t < hit.t || i > hit.prim) {
            hit.t = t; hit.prim = i; hit.b1 = b1; hit.b2 = b2;
          }
        }
        if (ANY && occluded) {
          cur = kRefEmpty;
          pend = kRefEmpty;
        } else {
          pend = last ? kRefEmpty : ~(i + 1);
        }
      }
#else
      unsigned long long m_inner = __ballot(cur >= 0);
      unsigned long long m_leaf = __ballot(cur < 0 && cur != kRefEmpty);
      int n_inner = __popcll(m_inner), n_leaf = __popcll(m_leaf);
      int n_free = kWave - n_inner - n_leaf;
      if (n_inner + n_leaf == 0) break;
      if (more && n_free >= kRefillIdle) break;
      if (n_inner * CASTER_VOTE_NUM >= n_leaf * CASTER_VOTE_DEN) {
        if (cur >= 0) cur = node_step<!ANY || !CASTER_ANY_UNSORTED>(S, cur, o, inv, hit.t, st, lane, overflow, s_top, n_top);
      } else if (cur < 0 && cur != kRefEmpty) {
        int i = ~cur;
        float t, b1, b2;
        bool last;
        bool found = prim_test(S, i, o, d, maxD, tmin, hit.t, t, b1, b2, last);
        if (found) {
          if (ANY) {
            if (t < hit.t || i > hit.prim) occluded = true;
          } else if (t < hit.t || i > hit.prim) {
            hit.t = t; hit.prim = i; hit.b1 = b1; hit.b2 = b2;
          }
        }
        if (ANY && occluded) cur = kRefEmpty;
        else cur = last ? stack_next(st, lane, hit.t) : ~(i + 1);
      }
#endif
    }
  }
#ifdef CASTER_WALK_STATS
  if (lane == 0)
    for (int k = 0; k < 8; ++k) atomicAdd(&g_walk_stats[ANY ? 1 : 0][k], ws[k]);
#endif
}

}  // namespace caster
