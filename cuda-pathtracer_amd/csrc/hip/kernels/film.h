// Part of the gfx950 wavefront renderer (included by ../kernels.hip, in this order: common.h, traverse.h, trace.h, shading.h, bsdf.h, path.h, volpath.h, lighttrace.h, bdpt.h, volpath_queues.h, film.h).
// Film accumulation and read-out, statistics, queue bookkeeping.
#pragma once

namespace caster {

// ----------------------------------------------------------------- film
// Ends every path of the batch and accumulates the film as Output() does
// (pathtracer.cu:2502-2513): acc += colour of iteration first..first+n-1, in
// iteration order.  A NaN/Inf sample adds nothing.  One thread per pixel, so
// the sum order is fixed.  Only iterations k0 .. k1-1 of the batch's n_iter are
// added (all of them, but for a batch that was traced ahead of the calls that
// ask for its iterations one by one).
__global__ void k_finish(LaunchCtx L, float* acc, uint32_t n_iter, uint32_t k0, uint32_t k1) {
  const DevScene& S = L.S;
  uint32_t npix = (uint32_t)S.width * (uint32_t)(S.row_y1 - S.row_y0);
  uint32_t invalid = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
    uint32_t px = i % (uint32_t)S.width, py = (uint32_t)S.row_y0 + i / (uint32_t)S.width;
    if (!row_live(S, py)) continue;
    uint32_t local = pixel_slot(S, px, py);
    uint32_t pixel = px + py * (uint32_t)S.width;
    V3 a = v3(acc[3 * pixel], acc[3 * pixel + 1], acc[3 * pixel + 2]);
    for (uint32_t k = k0; k < k1; ++k) {
      uint32_t slot = batch_slot(S, k, local, n_iter);
      float4 lf = L.P.li_flags[slot];
      V3 Li = v3(lf);
      uint32_t flags = __float_as_uint(lf.w);
      if (flags & kFPending) Li = S.integrator == 2 ? resolve_pending_vpt(L.P, (int)slot, Li, flags) : resolve_pending(L.P, (int)slot, Li, flags);
      bool bad = S.integrator == 0 ? has_nan(Li) : (has_nan(Li) || has_inf(Li));
      if (bad) ++invalid;
      else a = a + Li;
    }
    acc[3 * pixel] = a.x; acc[3 * pixel + 1] = a.y; acc[3 * pixel + 2] = a.z;
  }
  if (invalid) atomicAdd(&L.stats->invalid, (unsigned long long)invalid);
}

// out = tonemap(acc / iter) (pathtracer.cu:196-213, 2512-2516); mode 0 keeps
// the linear mean.
__global__ void k_resolve_film(const float* acc, float* out, uint32_t n_pix, float iter, int mode, int filmic) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pix; i += gridDim.x * blockDim.x) {
    V3 c = v3(acc[3 * i], acc[3 * i + 1], acc[3 * i + 2]) / iter;
    if (mode == 1) {
      if (filmic) {
        c = c - v3(0.004f, 0.004f, 0.004f);
        c = v3(fps::fmax_(0.f, c.x), fps::fmax_(0.f, c.y), fps::fmax_(0.f, c.z));
        c = (c * (6.2f * c + 0.5f)) / (c * (6.2f * c + 1.7f) + 0.06f);
      } else {
        float g = 1.f / 2.2f, e = 1.41421356f;
        c = v3(fps::fmax_(c.x, 1e-5f), fps::fmax_(c.y, 1e-5f), fps::fmax_(c.z, 1e-5f));
        c = v3(fps::pow_(c.x * e, g), fps::pow_(c.y * e, g), fps::pow_(c.z * e, g));
      }
    }
    out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
  }
}

// Primary-visibility map: primitive id per pixel from the traced camera queue.
__global__ void k_gather_ids(LaunchCtx L, RayQueue q, int32_t* ids) {
  const DevScene& S = L.S;
  uint32_t npix = (uint32_t)S.width * (uint32_t)S.height;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
    uint32_t px = i % (uint32_t)S.width, py = i / (uint32_t)S.width;
    ids[i] = __float_as_int(q.hit[pixel_slot(S, px, py)].y);
  }
}

// Adds the three queue lengths of a finished bounce to the ray statistics and
// clears the queues that the next shade pass appends to.
__global__ void k_roll(Stats* st, const uint32_t* n_ext, const uint32_t* n_shadow, const uint32_t* n_probe,
                       uint32_t* clear0, uint32_t* clear1, uint32_t* clear2, uint32_t ext_override, uint32_t* work,
                       uint32_t n_work, uint32_t work_init, uint32_t probes_in_shadow) {
  // work counters start at zero (k_trace adds the size of its static assignment itself)
  for (uint32_t i = 0; i < n_work; ++i) work[i * kWorkPad] = work_init;
  if (ext_override) st->rays[0] += ext_override;
  else if (n_ext) st->rays[0] += *n_ext;
  // (probe rays traced among the shadow rays are counted as probe rays)
  if (n_shadow) st->rays[1] += *n_shadow - (probes_in_shadow && n_probe ? *n_probe : 0u);
  if (n_probe) st->rays[2] += *n_probe;
  if (clear0) *clear0 = 0;
  if (clear1) *clear1 = 0;
  if (clear2) *clear2 = 0;
}

}  // namespace caster
