// Part of the gfx950 wavefront renderer (included by ../kernels.hip last).
// Parity hooks: the device functions the shade stage is made of, run on inputs of the caller's, one item per
// thread, so that a test can compare each with the CPU oracle's function item by item (in a film they are only
// seen through whole paths).  Input and output records are rows of floats; integers travel as their bit patterns.
#pragma once

namespace caster {

// in: material index | wo.xyz | n.xyz | dpdu.xyz | uv.xy | u.xyz  (16 floats, the last unused)
// out: wi.xyz | fr.xyz | pdf  (7 floats) -- SampleBSDF (pathtracer.cu:500-698), radiance transport
__global__ void k_hook_sample_bsdf(DevScene S, uint32_t n, const float* in, float* out) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float* r = in + 16 * (size_t)i;
    Mat m;
    load_mat(S, __float_as_int(r[0]), m);
    V3 wi, fr;
    float pdf;
    sample_bsdf(S, m, v3(r + 1), v3(r + 4), V2{r[10], r[11]}, v3(r + 7), v3(r + 12), wi, fr, pdf);
    float* o = out + 7 * (size_t)i;
    o[0] = wi.x; o[1] = wi.y; o[2] = wi.z; o[3] = fr.x; o[4] = fr.y; o[5] = fr.z; o[6] = pdf;
  }
}
// in: material index | wo.xyz | wi.xyz | n.xyz | dpdu.xyz | uv.xy  (16 floats, the last unused)
// out: fr.xyz | pdf  (4 floats) -- Fr (pathtracer.cu:700-818)
__global__ void k_hook_eval_bsdf(DevScene S, uint32_t n, const float* in, float* out) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float* r = in + 16 * (size_t)i;
    Mat m;
    load_mat(S, __float_as_int(r[0]), m);
    V3 fr;
    float pdf;
    eval_bsdf(S, m, v3(r + 1), v3(r + 4), v3(r + 7), V2{r[13], r[14]}, v3(r + 10), fr, pdf);
    float* o = out + 4 * (size_t)i;
    o[0] = fr.x; o[1] = fr.y; o[2] = fr.z; o[3] = pdf;
  }
}
// in: emitter index | pos.xyz | u.xy  (6 floats)
// out: radiance.xyz | dir.xyz | tmax | normal.xyz | pdf  (11 floats) -- Area::SampleLight (area.h:12-17)
__global__ void k_hook_sample_light(DevScene S, uint32_t n, const float* in, float* out) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float* r = in + 6 * (size_t)i;
    V3 rad, dir, nor;
    float tmax, pdf;
    sample_area_light(S, __float_as_int(r[0]), v3(r + 1), r[4], r[5], rad, dir, tmax, pdf, nor);
    float* o = out + 11 * (size_t)i;
    o[0] = rad.x; o[1] = rad.y; o[2] = rad.z; o[3] = dir.x; o[4] = dir.y; o[5] = dir.z; o[6] = tmax;
    o[7] = nor.x; o[8] = nor.y; o[9] = nor.z; o[10] = pdf;
  }
}
// in: x | y | iteration  (3 integers); out: origin.xyz | dir.xyz -- the camera sample of Path() (pathtracer.cu:881-890,
// Camera.h:45-85): seed, two jitter and two lens numbers, primary ray
__global__ void k_hook_camera_ray(DevScene S, uint32_t n, const float* in, float* out) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float* r = in + 3 * (size_t)i;
    uint32_t px = __float_as_uint(r[0]), py = __float_as_uint(r[1]), iter = __float_as_uint(r[2]);
    uint32_t rng = rng_seed(px + py * (uint32_t)S.width, iter);
    float ox = rng_uniform(rng) - 0.5f;
    float oy = rng_uniform(rng) - 0.5f;
    float a1 = rng_uniform(rng);
    float a2 = rng_uniform(rng);
    V3 orig, dir;
    primary_ray(S.cam, px + ox, py + oy, a1, a2, orig, dir);
    float* o = out + 6 * (size_t)i;
    o[0] = orig.x; o[1] = orig.y; o[2] = orig.z; o[3] = dir.x; o[4] = dir.y; o[5] = dir.z;
  }
}
// in: origin.xyz | dir.xyz | t | primitive  (8 floats); out: pos.xyz | n.xyz | uv.xy | dpdu.xyz | material, emitter,
// medium inside, medium outside  (15 floats) -- the intersection record of a hit as the shade stage rebuilds it
// (Triangle.h:121-141, sphere.h:69-88, line.h:74-84); b1, b2 are found again with the primitive's own test
__global__ void k_hook_hit_record(DevScene S, uint32_t n, const float* in, float* out) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float* r = in + 8 * (size_t)i;
    V3 o = v3(r), d = v3(r + 3);
    Hit h;
    h.t = r[6]; h.prim = __float_as_int(r[7]); h.b1 = h.b2 = 0.f;
    float t, b1, b2;
    bool last;
    if (prim_test(S, h.prim, o, d, maxabs3(d), S.epsilon, INFINITY, t, b1, b2, last)) { h.b1 = b1; h.b2 = b2; }
    Isect is;
    load_isect(S, o, d, h, is);
    float* q = out + 15 * (size_t)i;
    q[0] = is.pos.x; q[1] = is.pos.y; q[2] = is.pos.z; q[3] = is.n.x; q[4] = is.n.y; q[5] = is.n.z;
    q[6] = is.uv.x; q[7] = is.uv.y; q[8] = is.dpdu.x; q[9] = is.dpdu.y; q[10] = is.dpdu.z;
    q[11] = __int_as_float(is.mat); q[12] = __int_as_float(is.light); q[13] = __int_as_float(is.med_in);
    q[14] = __int_as_float(is.med_out);
  }
}

}  // namespace caster
