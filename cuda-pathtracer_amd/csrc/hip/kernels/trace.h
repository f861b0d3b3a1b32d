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
__global__ __launch_bounds__(kWave) void k_generate(LaunchCtx L, RayQueue q, uint32_t first_iter, uint32_t n_iter) {
  const DevScene& S = L.S;
  uint32_t total = (uint32_t)S.slots_per_iter * n_iter;
  for (uint32_t idx = blockIdx.x * kWave + threadIdx.x; idx < total; idx += gridDim.x * kWave) {
    uint32_t k, local;
    batch_slot_split(S, idx, n_iter, k, local);
    uint32_t px, py;
    bool live = slot_pixel(S, local, px, py);
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
// A wavefront claims rays a chunk at a time from the queue's work counter and keeps its lanes fed:
// when kRefillIdle of its ray slots are free they take the next unclaimed rays, so a long ray does
// not hold the others hostage.  Between refills every step does the kind of work (a 4-wide node, or
// one primitive test) that more lanes are waiting for.
//
// With CASTER_TWO_RAYS every lane holds two rays and takes part in a step when either of them can
// (both live in registers; a step picks the operands of the one it advances with selects and puts
// what changed back the same way).  The kernel is bound by vector instruction issue, and with one
// ray per lane a step finds some 42 of 64 lanes with the kind of work the wavefront voted for: a ray
// that waits for its last leaf sits through the node steps, one still descending through the
// primitive steps.  Measured on the benchmark scene: 50 lanes per node step instead of 42, 16 %
// fewer node steps and 14 % fewer primitive steps -- and launches 12 % LONGER (3.07 against 2.75
// ms): the rays that can always advance finish sooner, so the lanes fill with rays that wait; the
// selects add a tenth to every step; and 114 registers leave four wavefronts per SIMD, not five.
// Exchanging the two register sets instead (v_swap_b32 under a narrowed execution mask) made the
// compiler hold both sets in flight everywhere: 165 registers.  The default is one ray per lane.
//   ANY=false: closest hit; (t, prim, b1, b2) goes into the queue entry.
//   ANY=true:  any hit; an occluded ray clears the pending light-sampled term
//              of its path.
#ifndef CASTER_OCTANT_BOXES
#define CASTER_OCTANT_BOXES 1   // 1: the node step reads a ray's near and far planes by its direction octant (box_test_octant)
#endif
#ifndef CASTER_OCTANT_GUARD
#define CASTER_OCTANT_GUARD 1   // 0 (to see the test of axis-parallel rays fail): no ray keeps to the plain slab test
#endif
#ifndef CASTER_ANY_UNSORTED
#define CASTER_ANY_UNSORTED 1
#endif
#ifndef CASTER_TRACE_WAVES
#define CASTER_TRACE_WAVES (CASTER_TWO_RAYS ? 4 : 5)
#endif
constexpr int kTraceRays = CASTER_TWO_RAYS ? 2 : 1;   // rays per lane
#ifndef CASTER_REFILL_ANY
#define CASTER_REFILL_ANY CASTER_REFILL
#endif
#ifdef CASTER_WALK_STATS
// tuning aid: wavefront-level step counts of k_trace, [ANY][counter]
__device__ unsigned long long g_walk_stats[2][16];
#define CASTER_WS(i, v) ws[i] += (unsigned long long)(v)
#else
#define CASTER_WS(i, v)
#endif
#ifdef CASTER_DEBUG_SLOW_RAYS
// diagnosis builds only: rays that take more than CASTER_DEBUG_SLOW_RAYS steps are recorded here
__device__ unsigned int g_slow_count;
__device__ float g_slow[256][12];
#endif
// One ray of a lane.
#ifndef CASTER_WALK_KEEP_NOI
#define CASTER_WALK_KEEP_NOI 1   // 1: a ray keeps -(origin / direction) in registers; 0: three multiplies in every node step
#endif
struct WalkRay {
  V3 o, d, inv;
#if CASTER_WALK_KEEP_NOI
  V3 noi;    // -(o * inv), the addend of the fused slab test
#endif
  float maxD;
  Hit hit;
  int idx;   // queue entry; -1: the slot is free
  int cur;   // where the ray stands: a node (>= 0), a leaf (~index of its next primitive), kRefEmpty
  int pend;  // a leaf waiting for its primitives to be tested (same encoding), kRefEmpty
  int sp;    // height of its stack
  int occluded;
#ifdef CASTER_DEBUG_SLOW_RAYS
  unsigned int steps = 0;
#endif
  Octant oct;  // direction octant (node_step<.., OCT>); kOctPlain where origin / direction is not finite in
               // some component: the ray keeps to the plain slab test in the reference's form
};
// What the slab test wants of ray `r`.
__device__ inline SlabRay slab_of(const WalkRay& r) {
  SlabRay s;
  s.o = r.o; s.inv = r.inv;
#if CASTER_WALK_KEEP_NOI
  s.noi = r.noi;
#else
  s.noi = v3(-(r.o.x * r.inv.x), -(r.o.y * r.inv.y), -(r.o.z * r.inv.z));
#endif
  s.fused = (r.oct.b & kOctPlain) == 0u;
  return s;
}
__device__ inline void walk_ray_idle(WalkRay& r) {
  r.o = v3(0, 0, 0); r.d = v3(0, 0, 1); r.inv = v3(0, 0, 1); r.maxD = 1.f;
#if CASTER_WALK_KEEP_NOI
  r.noi = v3(0, 0, 0);
#endif
  r.hit.t = 0.f; r.hit.prim = -1; r.hit.b1 = r.hit.b2 = 0.f;
  r.idx = -1; r.cur = kRefEmpty; r.pend = kRefEmpty; r.sp = 0; r.occluded = 0;
  r.oct = Octant{0u, 0u};
}
__device__ inline float pick(bool second, float x0, float x1) { return second ? x1 : x0; }
__device__ inline int pick(bool second, int x0, int x1) { return second ? x1 : x0; }
__device__ inline V3 pick(bool second, V3 x0, V3 x1) { return v3(pick(second, x0.x, x1.x), pick(second, x0.y, x1.y), pick(second, x0.z, x1.z)); }
__device__ inline void put(bool second, float& x0, float& x1, float v) { x0 = second ? x0 : v; x1 = second ? v : x1; }
__device__ inline void put(bool second, int& x0, int& x1, int v) { x0 = second ? x0 : v; x1 = second ? v : x1; }

// (registers: the budget of the wavefronts the grid brings, 96 at five per SIMD.  With the fused slab test a ray
// carries three registers more, and at 80 -- room for a sixth wavefront, of another stream's shade kernel -- five
// values were spilled and read back inside the step loop: 1735 against 1830 Msamples/s on the benchmark scene)
#ifndef CASTER_TRACE_REG_WAVES
#define CASTER_TRACE_REG_WAVES CASTER_TRACE_WAVES
#endif
template <bool ANY>
__global__ __launch_bounds__(kWave * kTraceBlockWaves, CASTER_TRACE_REG_WAVES) void k_trace(LaunchCtx L, RayQueue q, WalkAux aux) {
  __shared__ uint2 s_stack[kTraceBlockWaves][kTraceRays * kLdsDepth][kWave];
  __shared__ float4 s_top[kTopNodes > 0 ? kTopNodes * kTopF4 : 1];
#ifdef CASTER_WALK_STATS
  unsigned long long ws[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
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
  // (free slots refilled at; an any-hit ray ends abruptly, so its wavefronts run emptier between refills)
  constexpr int kRefill = ANY ? CASTER_REFILL_ANY : kRefillIdle;
  Stack st{s_stack[wave_in_block], aux.spill + (wave_id * kWave + lane), aux.spill_stride, 0, 0};
  WalkRay a, b;   // the lane's two rays; a's stack is stack 0, b's stack 1
  walk_ray_idle(a);
  walk_ray_idle(b);
  const float tmin = S.epsilon;
  // node records fetched from memory and primitive records tested, for the statistics
  uint32_t n_node_fetches = 0, n_prim_tests = 0;
  // this wavefront's first chunk is assigned statically (no atomic storm at
  // launch): the work counters start at the end of the static assignment
  const uint32_t part = ((count + kWorkShards - 1) / kWorkShards + kWave - 1) / kWave * kWave;
  // Rays are claimed kChunk at a time from a short queue (so that they spread over all wavefronts)
  // and up to four times as many at a time from a long one (fewer trips to the work counters).
  const uint32_t n_waves = gridDim.x * kTraceBlockWaves;
  const uint32_t per_wave = count / (n_waves * kChunk);
  const uint32_t chunk = per_wave >= 16 ? 4 * kChunk : per_wave >= 8 ? 2 * kChunk : kChunk;
  uint32_t shard = wave_id % kWorkShards, shards_tried = 0;
  uint32_t chunk_next, chunk_end;
  {
    uint32_t lo = shard * part, hi = lo + part < count ? lo + part : count;
    uint32_t base = (wave_id / kWorkShards) * chunk;
    chunk_next = lo + base < hi ? lo + base : 0;
    chunk_end = lo + base < hi ? (chunk_next + chunk < hi ? chunk_next + chunk : hi) : 0;
  }
  bool more = true;
  // Writes out ray `r` where it is finished.
  auto retire = [&](WalkRay& r) {
    if (r.idx >= 0 && r.cur == kRefEmpty && r.pend == kRefEmpty) {
#ifdef CASTER_DEBUG_SLOW_RAYS
      if (r.steps > (unsigned)CASTER_DEBUG_SLOW_RAYS) {
        unsigned int k = atomicAdd(&g_slow_count, 1u);
        if (k < 256u) {
          float4 o4 = q.o[r.idx], d4 = q.d[r.idx];
          float* g = g_slow[k];
          g[0] = o4.x; g[1] = o4.y; g[2] = o4.z; g[3] = o4.w; g[4] = d4.x; g[5] = d4.y; g[6] = d4.z; g[7] = d4.w;
          g[8] = (float)r.steps; g[9] = ANY ? 1.f : 0.f; g[10] = r.hit.t; g[11] = __int_as_float(r.hit.prim);
        }
      }
#endif
      if (ANY) {
        if (r.occluded) {
          int slot = q.slot[r.idx];
          if (r.hit.prim < 0) {
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
        q.hit[r.idx] = make_float4(r.hit.t, __int_as_float(r.hit.prim), r.hit.b1, r.hit.b2);
      }
      r.idx = -1;
    }
  };
  // Gives the lanes whose slot `r` is free the next unclaimed rays.
  auto refill = [&](WalkRay& r) {
    unsigned long long idle_mask = lane_ballot(r.idx < 0);
    uint32_t n_idle = (uint32_t)__popcll(idle_mask);
    if (!more || n_idle == 0) return;
    if (chunk_next == chunk_end) {
      // Claim the next chunk.  The queue is cut into kWorkShards contiguous
      // parts with a work counter each: same-address atomics with a return
      // value retire at only ~15 per microsecond on this chip, so one shared
      // counter would bound the kernel.  A wavefront works on "its" part
      // first and moves on to the next part when that one is exhausted.
      while (shards_tried < kWorkShards) {
        uint32_t lo = shard * part, hi = lo + part < count ? lo + part : count;
        // the counter of a part counts from the end of what was handed out statically: one chunk
        // to each of the wavefronts whose first part it is
        const uint32_t work_init = (n_waves - shard + kWorkShards - 1) / kWorkShards * chunk;
        // A part whose rays were all handed out statically needs no atomic to be found empty: in a
        // launch over a short queue that is every part, and thousands of wavefronts asking eight
        // counters in turn would take longer than the tracing (same-address atomics retire at ~20 per us).
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
      if (shards_tried == kWorkShards) { more = false; return; }
    }
    uint32_t avail = chunk_end - chunk_next;
    uint32_t give = avail < n_idle ? avail : n_idle;
    uint32_t prefix = (uint32_t)__popcll(idle_mask & ((1ull << lane) - 1ull));
    if (r.idx < 0 && prefix < give) {
      r.idx = (int)(chunk_next + prefix);
      float4 o4 = q.o[r.idx], d4 = q.d[r.idx];
      r.o = v3(o4); r.d = v3(d4);
      r.inv = v3(1.f / r.d.x, 1.f / r.d.y, 1.f / r.d.z);
      r.maxD = maxabs3(r.d);
      const V3 noi = v3(-(r.o.x * r.inv.x), -(r.o.y * r.inv.y), -(r.o.z * r.inv.z));
#if CASTER_WALK_KEEP_NOI
      r.noi = noi;
#endif
      r.oct = octant_of(r.inv, CASTER_OCTANT_GUARD && !fps::slab_fused_(noi.x, noi.y, noi.z));
      // every queued ray starts at the scene's epsilon; o.w carries a hit known
      // beforehand (a probe ray's nearest emitter, at distance tmax) or -1
      // (no hit is looked for beyond the scene's far distance; a bound that is not a number gives way to it too)
      r.hit.t = d4.w < S.far_t ? d4.w : S.far_t; r.hit.prim = __float_as_int(o4.w); r.hit.b1 = r.hit.b2 = 0.f;
      r.occluded = 0;
#ifdef CASTER_DEBUG_SLOW_RAYS
      r.steps = 0;
#endif
      r.pend = kRefEmpty;
      r.sp = 0;
      // tmax < 0 marks a camera slot outside the image: nothing to trace.  (A tmax that is not
      // a number -- a light sample at the vertex itself -- bounds nothing in the reference: every
      // comparison with it fails, in the slab test and in the primitive tests; here such a ray is
      // traced as far as the scene's far distance, like a ray without a bound.)
      r.cur = (!(d4.w < 0.f) && S.n_nodes) ? 0 : kRefEmpty;
    }
    chunk_next += give;
  };
  for (;;) {
    // finished rays leave
    retire(a);
    if (kTraceRays == 2) retire(b);
    // free slots are refilled once there are enough of them
    uint32_t n_idle = (uint32_t)__popcll(lane_ballot(a.idx < 0)) + (kTraceRays == 2 ? (uint32_t)__popcll(lane_ballot(b.idx < 0)) : 0u);
    if (more && n_idle >= (uint32_t)kRefill) {
      refill(a);
      if (kTraceRays == 2) refill(b);
    }
    CASTER_WS(6, 1);
    if (!any_lane(a.idx >= 0 || b.idx >= 0)) {
      if (!more) break;
      continue;
    }
    // Advance until enough slots are free again.  Each step the wavefront does whichever kind of
    // work more of its lanes can do: one 4-wide node for every lane with a ray that stands at an
    // inner node, or one primitive test for every lane with a ray that holds a leaf.  A ray that
    // stands at a leaf parks it in `pend` (a leaf reference is ~index of the next primitive to test,
    // so the cursor lives in the reference itself) and goes on with its next deferred node, so that
    // it keeps taking part in node steps; the parked leaf may later shorten the ray and make some of
    // that node work unnecessary, never wrong.  Every ray is kept in that state: one that stands at
    // a leaf has a parked leaf already.
    for (;;) {
      const bool node_a = a.cur >= 0, node_b = kTraceRays == 2 && b.cur >= 0;
      const bool leaf_a = a.pend != kRefEmpty, leaf_b = kTraceRays == 2 && b.pend != kRefEmpty;
      const int n_inner = __popcll(lane_ballot(node_a || node_b)), n_leaf = __popcll(lane_ballot(leaf_a || leaf_b));
      // (votes on single compares and the masks combined as scalars: the vote on an `or` of two flags is made
      // a number per lane and compared again)
      const int n_busy = __popcll(lane_ballot(a.cur != kRefEmpty) | lane_ballot(a.pend != kRefEmpty)) +
                         (kTraceRays == 2 ? __popcll(lane_ballot(b.cur != kRefEmpty) | lane_ballot(b.pend != kRefEmpty)) : 0);
      if (n_busy == 0) break;
      if (more && kTraceRays * kWave - n_busy >= kRefill) break;
      CASTER_WS(4, n_busy);
#ifdef CASTER_DEBUG_SLOW_RAYS
      if (a.idx >= 0 && (a.cur != kRefEmpty || a.pend != kRefEmpty)) ++a.steps;
#endif
      if (n_inner * CASTER_VOTE_NUM >= n_leaf * CASTER_VOTE_DEN) {
        CASTER_WS(0, 1);
        CASTER_WS(1, n_inner);
        CASTER_WS(5, any_lane(a.cur >= 0 && a.sp + 3 > kLdsDepth) ? 1 : 0);
        CASTER_WS(7, __popcll(lane_ballot(a.cur >= 0)) * 0 + __popcll(lane_ballot(a.cur >= 0 && a.sp + 3 > kLdsDepth)));
        // the lane's first ray if it stands at a node, else its second
        const bool w = kTraceRays == 2 && !node_a;
        int cur = pick(w, a.cur, b.cur);
        n_node_fetches += (uint32_t)__popcll(lane_ballot(cur >= n_top));
        // (one ray in 10^7 is parallel to an axis; the whole step then takes the plain slab test)
        const bool plain_step = !CASTER_OCTANT_BOXES || any_lane(cur >= 0 && ((kTraceRays == 2 ? (a.oct.b | b.oct.b) : a.oct.b) & kOctPlain) != 0);
        if (cur >= 0) {
          const SlabRay sr = slab_of(kTraceRays == 2 && w ? b : a);
          const float tmax = pick(w, a.hit.t, b.hit.t);
          int pend = pick(w, a.pend, b.pend);
          st.sp = pick(w, a.sp, b.sp);
          st.ctx = w ? 1 : 0;
          if (plain_step) cur = node_step<!ANY || !CASTER_ANY_UNSORTED, false>(S, cur, sr, tmax, st, lane, overflow, s_top, n_top);
          else cur = node_step<!ANY || !CASTER_ANY_UNSORTED, true>(S, cur, sr, tmax, st, lane, overflow, s_top, n_top,
                                                                   kTraceRays == 2 && w ? b.oct : a.oct);
          if (cur < 0 && cur != kRefEmpty && pend == kRefEmpty) {
            pend = cur;
            cur = stack_next_of<!ANY || !CASTER_ANY_UNSORTED>(st, lane, tmax);
          }
          put(w, a.cur, b.cur, cur);
          put(w, a.pend, b.pend, pend);
          put(w, a.sp, b.sp, st.sp);
        }
      } else {
        CASTER_WS(2, 1);
        CASTER_WS(3, n_leaf);
        const bool w = kTraceRays == 2 && !leaf_a;
        int pend = pick(w, a.pend, b.pend);
        n_prim_tests += (uint32_t)n_leaf;   // (every lane that holds a leaf tests one primitive)
#ifdef CASTER_WALK_STATS
        bool ws_last = true;
#endif
        if (pend != kRefEmpty) {
          const SlabRay sr = slab_of(kTraceRays == 2 && w ? b : a);
          const V3 d = pick(w, a.d, b.d);
          const float maxD = kTraceRays == 2 ? pick(w, a.maxD, b.maxD) : maxabs3(d);   // (one v_max3 against a register held)
          float ht = pick(w, a.hit.t, b.hit.t);
          const int hp = pick(w, a.hit.prim, b.hit.prim);
          int cur = pick(w, a.cur, b.cur);
          int i = ~pend;
          float t, b1, b2;
          bool last;
          bool found = prim_test(S, i, sr, d, maxD, tmin, ht, t, b1, b2, last);
#ifdef CASTER_WALK_STATS
          ws_last = last;
#endif
          // (an occlusion ray holds -1 in hit.prim, or the emitter a probe ray was sent to, at
          // distance hit.t: the emitter itself and what it hides at equal distance do not occlude)
          const bool better = found && (t < ht || i > hp);
          if (ANY) {
            if (better) {
              put(w, a.occluded, b.occluded, 1);
              cur = kRefEmpty;
              pend = kRefEmpty;
            } else {
              pend = last ? kRefEmpty : ~(i + 1);
            }
          } else {
            if (better) {
              ht = t;
              put(w, a.hit.t, b.hit.t, t); put(w, a.hit.prim, b.hit.prim, i);
              put(w, a.hit.b1, b.hit.b1, b1); put(w, a.hit.b2, b.hit.b2, b2);
            }
            pend = last ? kRefEmpty : ~(i + 1);
          }
          if (cur < 0 && cur != kRefEmpty && pend == kRefEmpty) {   // the ray stands at another leaf already
            st.sp = pick(w, a.sp, b.sp);
            st.ctx = w ? 1 : 0;
            pend = cur;
            cur = stack_next_of<!ANY || !CASTER_ANY_UNSORTED>(st, lane, ht);
            put(w, a.sp, b.sp, st.sp);
          }
          put(w, a.cur, b.cur, cur);
          put(w, a.pend, b.pend, pend);
        }
#ifdef CASTER_WALK_STATS
        {   // what idle lanes could take over: the next primitive of a lane whose leaf goes on
          const int n_want = __popcll(lane_ballot(!ws_last));
          const int n_free = kWave - n_leaf;
          ws[13] += (unsigned long long)n_want;
          ws[14] += (unsigned long long)(n_want < n_free ? n_want : n_free);
          ws[15] += (unsigned long long)n_free;
        }
#endif
      }
    }
  }
  if (n_node_fetches | n_prim_tests) {
    // (both are sums of wavefront-wide counts, the same in every lane)
    if (lane == 0) {
      atomicAdd(&L.stats->node_fetches[ANY ? 1 : 0], (unsigned long long)n_node_fetches);
      atomicAdd(&L.stats->prim_tests[ANY ? 1 : 0], (unsigned long long)n_prim_tests);
    }
  }
#ifdef CASTER_WALK_STATS
  if (lane == 0)
    for (int k = 0; k < 8; ++k) atomicAdd(&g_walk_stats[ANY ? 1 : 0][k], ws[k]);
  if (lane == 0)
    for (int k = 13; k < 16; ++k) atomicAdd(&g_walk_stats[ANY ? 1 : 0][k], ws[k]);
  atomicAdd(&g_walk_stats[ANY ? 1 : 0][8], (unsigned long long)st.n_pop_calls);
  atomicAdd(&g_walk_stats[ANY ? 1 : 0][9], (unsigned long long)st.n_pop_iters);
  atomicAdd(&g_walk_stats[ANY ? 1 : 0][10], (unsigned long long)st.n_spill_push);
  atomicAdd(&g_walk_stats[ANY ? 1 : 0][11], (unsigned long long)st.n_spill_pop);
  atomicAdd(&g_walk_stats[ANY ? 1 : 0][12], (unsigned long long)st.n_slow);
#endif
}

}  // namespace caster
