/* C-ABI of the MI355X render hot path (libcaster.so).
 *
 * Drop-in boundary for the reference's device runtime,
 *   void BeginRender(Scene&, unsigned w, unsigned h, float eps);
 *   void Render(Scene&, unsigned w, unsigned h, Camera*, unsigned iter,
 *               bool reset, float3* output);
 *   void EndRender();
 * (CUDA-PathTracer/tracer/pathtracer.h:8-10), and for the scene loading that
 * feeds it (InitScene, CUDA-PathTracer/tracer/parsescene.h:26-27).  Plain
 * pointers and sizes only; every call returns a status code and never throws.
 * One host thread per renderer handle.
 *
 * Image convention (as the reference's kernels write it,
 * CUDA-PathTracer/tracer/Camera.h:57-58): W*H*3 floats, RGB interleaved,
 * pixel = x + y*W with row 0 the BOTTOM scanline. */
#ifndef CASTER_H
#define CASTER_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct caster_scene caster_scene;
typedef struct caster_renderer caster_renderer;

enum {
  CASTER_OK = 0,
  CASTER_ERR_ARG = 1,      /* null / out-of-range argument */
  CASTER_ERR_SCENE = 2,    /* scene file could not be read or is malformed */
  CASTER_ERR_DEVICE = 3,   /* a HIP call failed */
  CASTER_ERR_OVERFLOW = 4  /* BVH deeper than the traversal stack: image invalid */
};

typedef struct caster_scene_info {
  int32_t width, height;
  int32_t integrator;      /* 0 ao, 1 pt, 2 vpt, 3 lt, 4 bdpt (IntegratorType, scene.h:13-22) */
  int32_t max_depth;
  int32_t n_prims, n_nodes, n_lights, n_materials, n_media, n_textures;
  int32_t has_environment;
  int32_t filmic;
  float max_dist, epsilon;
} caster_scene_info;

typedef struct caster_stats {
  /* rays traced since caster_begin / the last reset: 0 extension (camera +
     continuation), 1 shadow (any-hit), 2 BSDF-sampled light probe, 3
     transmittance segment */
  uint64_t rays[4];
  uint64_t samples;        /* camera samples = pixels * iterations rendered */
  uint64_t invalid;        /* samples dropped because their radiance was NaN/Inf */
  uint64_t stack_overflow; /* must stay 0 */
} caster_stats;

/* Replaces InitScene (parsescene.cpp:600-617): parses the JSON scene file,
   loads its meshes/textures, builds the BVH and the light distribution. */
int caster_scene_load(const char* json_path, caster_scene** out);
/* caster_scene_load with the BVH built on HIP device `device_id` instead of on the
   host (replaces BVH::build, bvh.cpp:8-180, which dominates the load time of
   large meshes).  The tree and the primitive order are those of the host build,
   bit for bit. */
int caster_scene_load_device(const char* json_path, int device_id, caster_scene** out);
/* Parity hook (no counterpart in the reference): the pre-order binary tree.
   boxes: n_nodes*6 floats (lo.xyz, hi.xyz); meta: n_nodes*4 ints (second child,
   leaf flag, first and last primitive of a leaf or -1); prim_push_ids: for every
   primitive in tree order, its index in the scene file's push order.  Any
   pointer may be null. */
int caster_scene_get_bvh(const caster_scene* s, float* boxes, int32_t* meta, int32_t* prim_push_ids);
void caster_scene_free(caster_scene* s);
int caster_scene_get_info(const caster_scene* s, caster_scene_info* out);

/* Replaces BeginRender (pathtracer.cu:2554-2681): uploads the scene to HIP
   device `device_id` and allocates the film and path pools.  The renderer
   keeps no reference to `s` afterwards.  `iters_in_flight` is how many
   iterations (samples per pixel) are traced together as one batch of paths;
   0 picks a default: 64 M paths per stream (32 iterations of 1920x1080, 76 GB of
   device memory with the default three streams) and never more than a quarter of
   the memory free on the device at that moment.  A stated value is taken as it
   is (twice the default is 1 % faster on a 1.4 M-triangle scene; a quarter of it
   6 % slower). */
int caster_begin(const caster_scene* s, int device_id, int iters_in_flight, caster_renderer** out);

/* caster_begin with the number of HIP streams stated.  Successive batches are
   dealt to the streams in turn and overlap on the GPU (each stream owns a path
   pool of `iters_in_flight` iterations; the film is still accumulated in
   iteration order, so the image does not depend on `n_streams`).  0 picks the
   default (3); at most 8.  No counterpart in the reference, whose Render
   synchronises the device after every sample (pathtracer.cu:2739). */
int caster_begin_streams(const caster_scene* s, int device_id, int iters_in_flight, int n_streams,
                         caster_renderer** out);

/* BeginRender for several GPUs behind one handle.  The reference is single-GPU
   (one process, the default stream, no peer access anywhere in tracer/); a sample
   depends on its pixel and iteration only (pathtracer.cu:881), so the image
   shards with no exchange until the film is read.  The scene is uploaded to
   every device of `device_ids` (a device may be named more than once: that many
   renderers then share it).  shard_mode CASTER_SHARD_TILE deals slabs of 8
   scanlines round-robin over the devices (every device renders every iteration
   of its own scanlines: the film is bit-identical to one device's);
   CASTER_SHARD_SPP deals blocks of iterations round-robin (every device renders
   the whole frame; the films differ from one device's by fp32 re-association).
   Every call on the handle drives all devices at once, each from a host thread
   of its own; caster_read_film / caster_read_film_device copy the other devices'
   films to the first device (hipMemcpyPeer) and add them up there.  The parity
   hooks (caster_primary_ids, caster_trace_rays*) run on the first device;
   caster_set_rows / caster_set_row_stripes are refused. */
enum { CASTER_SHARD_TILE = 0, CASTER_SHARD_SPP = 1 };
int caster_begin_multi(const caster_scene* s, const int* device_ids, int n_dev, int iters_in_flight, int n_streams,
                       int shard_mode, caster_renderer** out);

/* Restricts this renderer to scanlines [y0, y1) of the image (tile sharding
   across GPUs).  Default is the whole image. */
int caster_set_rows(caster_renderer* r, int y0, int y1);
/* Restricts this renderer to the scanlines y0 <= y < y1 with (y - y0) % stride < rows:
   slabs of `rows` scanlines every `stride` (interleaved tile sharding: GPU k of G
   takes (8k, height, 8, 8G), so that every GPU sees a sample of the whole image and
   none is left with its dearest band).  The reference launches one thread per pixel
   of the whole image (pathtracer.cu:2693-2703); a sample depends on its pixel and
   iteration only (pathtracer.cu:881), so any subset of scanlines renders the same
   values.  Where y0, rows and stride are multiples of 8 only those scanlines take
   path slots, and a batch holds proportionally more iterations.  Waits for work
   under way.  (1 <= rows <= stride.) */
int caster_set_row_stripes(caster_renderer* r, int y0, int y1, int rows, int stride);

/* Replaces n_iter calls of Render (pathtracer.cu:2691-2742) for iterations
   first_iter .. first_iter+n_iter-1 (the reference counts from 1,
   CudaRender.cpp:49): one sample per pixel per iteration, accumulated into the
   film in iteration order.  reset != 0 clears the film first (Output's
   `reset`, pathtracer.cu:2507).  Returns after the GPU has finished. */
int caster_render(caster_renderer* r, uint32_t first_iter, uint32_t n_iter, int reset);

/* Same, but only enqueues the work on the renderer's streams; pair with
   caster_sync.  For ao, pt and bdpt the call does not wait for the GPU.  For vpt
   and lt the number of rounds of a batch depends on the data (a boundary crossing
   consumes no bounce, pathtracer.cu:1106-1113), so after the first maxDepth
   rounds of every stream's batch have been enqueued the call waits for each
   batch in turn to ask whether rays are left: it returns once the last batch
   of the call has been enqueued in full, not before. */
int caster_render_async(caster_renderer* r, uint32_t first_iter, uint32_t n_iter, int reset);
int caster_sync(caster_renderer* r);

/* For the caller the reference has: CudaRender::renderImage asks for ONE iteration per frame, iteration
   numbers counting up (CudaRender.cpp:44-60), and a batch of one iteration leaves the GPU mostly idle.  With a
   look-ahead of n_iter > 1 a caster_render(r, i, 1, 0) call traces iterations i .. i+n_iter-1 together (at most
   the renderer's iterations in flight) and adds only iteration i to the film; the calls for i+1, i+2, ... then
   find their iteration traced and only add it, so the film after every call is what it would be without the
   look-ahead, bit for bit (a sample depends on its pixel and iteration alone, pathtracer.cu:881, and the film is
   accumulated in iteration order).  A call that does not continue the sequence (another iteration number, more
   than one iteration, reset, a change of the row set, a parity hook that uses the pools) drops what was traced
   ahead.  The ray counts of caster_get_stats include iterations traced ahead.  0 or 1 turns it off (the
   default).  Not for the splatting integrators (lt, bdpt), which add to the film while they trace. */
int caster_set_lookahead(caster_renderer* r, int n_iter);

/* Replaces Output + the D2H copy (pathtracer.cu:2502-2517,
   CudaRender.cpp:55).  mode 0: linear mean (sum / iterations rendered);
   mode 1: tonemapped as Output does (filmic or gamma per the scene's camera);
   mode 2: the raw accumulated sum.  `rgb` is host memory, W*H*3 floats. */
int caster_read_film(caster_renderer* r, float* rgb, int mode);

/* Same into DEVICE memory owned by the caller (the reference's `float3*
   output` is a device pointer too); used to hand the film to RCCL. */
int caster_read_film_device(caster_renderer* r, float* rgb_device, int mode);

/* Replaces ImageIO::SavePng / SaveExr (imageio.cpp:62-79, 105-163) as called by
   CudaRender::SaveImage (CudaRender.cpp:84-90).  The format follows the
   extension: .png = 8-bit RGB, top row first, unsigned(clamp(v,0,1)*255) as
   SavePng; .exr = half-float B,G,R scanlines, top row first; .pfm = raw float32,
   bottom row first.  caster_write_image writes caller-owned host pixels (W*H*3
   floats, row 0 = bottom scanline); caster_save_image reads the film first,
   with `mode` as in caster_read_film or -1 for the format's natural choice
   (png: tonemapped, exr/pfm: linear mean). */
int caster_write_image(const char* path, int width, int height, const float* rgb);
int caster_save_image(caster_renderer* r, const char* path, int mode);

/* Adds `n_iters` to the iteration count and `sum_device` (W*H*3 floats on this
   renderer's device) to the film: merges a film reduced from other GPUs. */
int caster_film_add_device(caster_renderer* r, const float* sum_device, uint32_t n_iters);

/* Film checkpoint and resume.  The reference keeps its accumulation buffer on the device only and
   has no way to continue a render (its one cache, bvh.cache, is read-only: bvh.cpp:196-224).  A sample
   depends on its pixel and iteration alone (pathtracer.cu:881) and the film is added up in iteration
   order, so the raw sum after n iterations plus n is all there is to keep: caster_checkpoint_save
   writes both (a 40-byte header -- "CSTRFILM", version, width, height, iterations, integrator -- then
   W*H*3 float32, row 0 = bottom scanline; written to `path`.part and renamed), caster_checkpoint_load
   replaces the renderer's film and iteration count with the file's, after which
   caster_render(r, caster_iterations(r) + 1, ...) continues the render: n + m iterations resumed equal
   n + m iterations in one go, bit for bit.  A file of another image size or integrator is refused. */
int caster_checkpoint_save(caster_renderer* r, const char* path);
int caster_checkpoint_load(caster_renderer* r, const char* path);

/* Parity hook (no counterpart in the reference): primitive id, in BVH leaf
   order, hit by the camera ray of every pixel for `iter`; -1 on a miss. */
int caster_primary_ids(caster_renderer* r, uint32_t iter, int32_t* ids);

/* Parity hook (no counterpart in the reference): the closest hit of `n` rays of
   the caller's, traced by the renderer's traversal kernel from the scene's
   epsilon to tmax[i].  origins and dirs hold n*3 floats; prim[i] receives the
   primitive id in BVH leaf order or -1, t[i] the distance.  n may not exceed
   the renderer's path slots (width*height*iterations in flight). */
int caster_trace_rays(caster_renderer* r, uint32_t n, const float* origins, const float* dirs, const float* tmax,
                      int32_t* prim, float* t);

/* Parity hook (no counterpart in the reference): one of the device functions the shade stage is made of, run on
   `n` input records of the caller's, one record per GPU thread; `in` and `out` are host arrays of n rows of floats
   (integers travel as their bit patterns).  Material and emitter indices must be valid for the scene.
     CASTER_FN_SAMPLE_BSDF   SampleBSDF (pathtracer.cu:500-698), radiance transport
                             in 16: material | wo.xyz | n.xyz | dpdu.xyz | uv.xy | u.xyz | -    out 7: wi.xyz | fr.xyz | pdf
     CASTER_FN_EVAL_BSDF     Fr (pathtracer.cu:700-818)
                             in 16: material | wo.xyz | wi.xyz | n.xyz | dpdu.xyz | uv.xy | -   out 4: fr.xyz | pdf
     CASTER_FN_SAMPLE_LIGHT  Area::SampleLight (area.h:12-17, Triangle.h:146-155)
                             in 6: emitter | pos.xyz | u.xy       out 11: radiance.xyz | dir.xyz | tmax | normal.xyz | pdf
     CASTER_FN_CAMERA_RAY    the camera sample of Path() (pathtracer.cu:881-890, Camera.h:45-85)
                             in 3: x | y | iteration (integers)   out 6: origin.xyz | dir.xyz
     CASTER_FN_HIT_RECORD    the Intersection record of a hit (Triangle.h:121-141, sphere.h:69-88, line.h:74-84)
                             in 8: origin.xyz | dir.xyz | t | primitive
                             out 15: pos.xyz | n.xyz | uv.xy | dpdu.xyz | material, emitter, medium inside, outside */
enum { CASTER_FN_SAMPLE_BSDF = 0, CASTER_FN_EVAL_BSDF, CASTER_FN_SAMPLE_LIGHT, CASTER_FN_CAMERA_RAY, CASTER_FN_HIT_RECORD,
       CASTER_N_FN };
int caster_eval_function(caster_renderer* r, int fn, uint32_t n, const float* in, float* out);

/* caster_trace_rays for callers that cannot allocate device memory (tight serving
   loops, harnesses that forbid hipMalloc, hosts whose allocator owns the card).  No
   counterpart in the reference.  The call waits for the renderer's work under way,
   launches on the renderer's first stream (whose stack-spill columns and statistics
   it borrows) and returns after that stream has drained: it cannot itself be
   captured into a HIP graph.  Every pointer is DEVICE memory on the renderer's
   GPU, laid out as in caster_trace_rays; the call makes no hipMalloc / hipFree (nor
   any other allocating call): the ray queue and the work counters its kernels need
   live in a static device array of CASTER_STATIC_MAX_RAYS * 52 bytes + 6.4 KB
   (13.6 MB) that every GPU the library runs on holds until the process ends.  n
   beyond CASTER_STATIC_MAX_RAYS is CASTER_ERR_ARG.  All calls on one GPU share
   that array: calls that overlap on one GPU (from several threads or renderers)
   race; consecutive calls do not.  Returns after the GPU has finished. */
#define CASTER_STATIC_MAX_RAYS (1u << 18)
int caster_trace_rays_static(caster_renderer* r, uint32_t n, const float* origins_dev, const float* dirs_dev,
                             const float* tmax_dev, int32_t* prim_dev, float* t_dev);

/* Per-kernel-class device time, measured with HIP events recorded on the
   renderer's own streams around every launch of the class's kernel (extend,
   shadow and probe each bracket exactly one traversal launch; with more than
   one stream a launch shares the GPU with the other stream's kernels while it
   runs).  No counterpart in the reference, which measures nothing.  Enabling
   resets the totals. */
enum {
  CASTER_K_GENERATE = 0, CASTER_K_EXTEND, CASTER_K_SHADE, CASTER_K_SHADOW, CASTER_K_PROBE, CASTER_K_FINISH,
  CASTER_N_KERNEL_CLASSES
};
typedef struct caster_kernel_times {
  double ms[CASTER_N_KERNEL_CLASSES];
  uint64_t launches[CASTER_N_KERNEL_CLASSES];
} caster_kernel_times;
int caster_set_profiling(caster_renderer* r, int enable);
int caster_get_kernel_times(caster_renderer* r, caster_kernel_times* out);

int caster_get_stats(caster_renderer* r, caster_stats* out);
/* What the traversal kernel itself fetched since caster_begin / the last reset (measurement aid,
   no counterpart in the reference): out4[0] = 4-wide node records read from memory by closest-hit
   rays (extension, probe, transmittance), out4[1] = primitive records they tested, out4[2] and
   out4[3] the same for any-hit (shadow) rays.  The top levels of the tree are served from LDS
   and not counted; walks done in place by the splatting integrators are not counted either. */
int caster_get_walk_counts(caster_renderer* r, uint64_t* out4);
int caster_reset_stats(caster_renderer* r);
uint32_t caster_iterations(const caster_renderer* r);
/* Iterations per batch of paths this renderer was given (the default depends on the
   resolution and on the free device memory). */
int caster_iters_in_flight(const caster_renderer* r);

/* Replaces EndRender (pathtracer.cu:2683-2689); frees everything. */
void caster_end(caster_renderer* r);

/* Message of the last failed call on this thread. */
const char* caster_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
