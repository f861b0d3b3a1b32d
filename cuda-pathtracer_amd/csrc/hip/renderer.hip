// Host side of the wavefront renderer and the C-ABI (include/caster.h).
//
// Replaces the reference's BeginRender / Render / EndRender
// (CUDA-PathTracer/tracer/pathtracer.cu:2554-2742): instead of one megakernel
// launch, one Output launch, a device synchronisation and a frame download per
// sample, a call traces `iters_in_flight` samples per pixel as one batch of
// paths through generate -> [extend -> shade -> shadow/probe]* -> finish on
// one of a few HIP streams with no host round trip inside the batch (queue
// lengths stay on the device), consecutive batches overlap on different
// streams, and the film stays resident until it is asked for.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/caster.h"
#include "../scene/scene.h"
#include "kernels.hip"

// Scratch of caster_trace_rays_static, the entry point that allocates nothing: the ray queue its
// kernels work on (origin | known hit, direction | tmax, slot, hit record), the queue length and the
// traversal kernel's work counters.  One copy per GPU, held until the process ends.  In host code
// this name is only a handle for hipGetSymbolAddress: the device address of each GPU's copy is
// looked up once per device (static_scratch_of) and that pointer is what kernels and memsets get.
namespace caster {
constexpr uint32_t kStaticMaxRays = CASTER_STATIC_MAX_RAYS;
constexpr size_t kStaticCounterWords = 64 + 3 * (size_t)kWorkShards * kWorkPad;
constexpr size_t kStaticScratchBytes = (size_t)kStaticMaxRays * (3 * sizeof(float4) + sizeof(int)) + kStaticCounterWords * sizeof(uint32_t);
}
static __device__ __align__(256) unsigned char scratch[caster::kStaticScratchBytes];

namespace caster {

static thread_local std::string g_last_error;
static thread_local bool g_out_of_memory = false;   // the calling thread's latest failed allocation ran out of device memory

static int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

#define HIPCHECK(expr)                                                                        \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return fail(CASTER_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));      \
  } while (0)

// Device address of `scratch` on each GPU (every device holds its own copy at its own address).
constexpr int kMaxDevices = 64;
static void* g_static_scratch[kMaxDevices] = {nullptr};
static std::mutex g_static_scratch_mutex;   // (caster_begin_multi starts one host thread per part, and parts may share a device)
// Looks the address up on first use for `device`, which must be the current device.
static int static_scratch_of(int device, void** out) {
  if (device < 0 || device >= kMaxDevices) return fail(CASTER_ERR_ARG, "device index beyond the static scratch table");
  std::lock_guard<std::mutex> lock(g_static_scratch_mutex);
  if (!g_static_scratch[device]) {
    void* p = nullptr;
    HIPCHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) return fail(CASTER_ERR_DEVICE, "hipGetSymbolAddress returned no address for the static scratch");
    g_static_scratch[device] = p;
  }
  *out = g_static_scratch[device];
  return CASTER_OK;
}

static inline float4 mk4(f3 v, float w) { return make_float4(v.x, v.y, v.z, w); }
static inline float as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline float as_float(uint32_t i) { float f; memcpy(&f, &i, 4); return f; }
static inline float maxabs(f3 v) { return fps::fmax_(fps::fmax_(fabsf(v.x), fabsf(v.y)), fabsf(v.z)); }

// Flat arrays in the device layout (device_scene.h), built once per scene.
struct HostArrays {
  std::vector<float4> nodes, isect, shade, lights, mats, media;
  std::vector<uint32_t> tex_pool;
  std::vector<int> tex_info;
  std::vector<float> env;
  std::vector<float> density;
};

static void build_arrays(const Scene& sc, HostArrays& A) {
  const size_t np = sc.prims.size();
  A.isect.assign(np * kIsectF4, make_float4(0, 0, 0, 0));
  A.shade.assign(np * kShadeF4, make_float4(0, 0, 0, 0));
  std::vector<uint32_t> tail(np, 0);
  for (size_t i = 0; i < np; ++i) tail[i] = (uint32_t)sc.prims[i].type & kPrimTypeMask;
  // The binary tree of the host build is collapsed into 4-wide nodes: a node's
  // child with the largest box that is itself an inner node is replaced by its
  // two children until four children are held.  A ray reaches a leaf exactly
  // when that leaf's own box test passes (a box that passes implies that every
  // enclosing box passes, because the slab test is monotone in the box), so
  // dropping intermediate boxes changes the work, not the result.  Leaf ranges
  // end at the primitive flagged "last".
  auto leaf_ref = [&](int host) {
    const Node& n = sc.nodes[host];
    if (n.start < 0) return kRefEmpty;
    tail[n.end] |= kPrimLastInLeaf;
    return ~n.start;
  };
  auto box_area = [&](int host) {
    f3 d = sc.nodes[host].hi - sc.nodes[host].lo;
    return d.x * d.y + d.y * d.z + d.z * d.x;
  };
  struct Work { int host; int dev; };
  std::vector<Work> todo;
  auto emit_node = [&]() { A.nodes.resize(A.nodes.size() + kNodeF4, make_float4(0, 0, 0, 0)); return (int)(A.nodes.size() / kNodeF4) - 1; };
  if (!sc.nodes.empty()) {
    int root_dev = emit_node();
    todo.push_back({0, root_dev});
    // depth-first emission keeps a subtree's nodes close together in memory
    while (!todo.empty()) {
      Work w = todo.back();
      todo.pop_back();
      int kids[4];
      int nk = 0;
      if (sc.nodes[w.host].leaf) {
        kids[nk++] = w.host;  // a scene whose root is a leaf
      } else {
        kids[nk++] = w.host + 1;
        kids[nk++] = sc.nodes[w.host].second;
        while (nk < 4) {
          int best = -1;
          float best_area = -1.f;
          for (int k = 0; k < nk; ++k)
            if (!sc.nodes[kids[k]].leaf && box_area(kids[k]) > best_area) { best_area = box_area(kids[k]); best = k; }
          if (best < 0) break;
          int h = kids[best];
          kids[best] = h + 1;
          kids[nk++] = sc.nodes[h].second;
        }
      }
      float lo[3][4], hi[3][4];
      int ref[4];
      for (int k = 0; k < 4; ++k) {
        if (k < nk) {
          const Node& c = sc.nodes[kids[k]];
          lo[0][k] = c.lo.x; lo[1][k] = c.lo.y; lo[2][k] = c.lo.z;
          hi[0][k] = c.hi.x; hi[1][k] = c.hi.y; hi[2][k] = c.hi.z;
          if (c.leaf) ref[k] = leaf_ref(kids[k]);
          else { ref[k] = emit_node(); todo.push_back({kids[k], ref[k]}); }
          // (a child without primitives gets the box of an empty slot: the octant slab test rejects a slot by
          // its box alone)
          if (ref[k] == kRefEmpty)
            for (int a = 0; a < 3; ++a) { lo[a][k] = INFINITY; hi[a][k] = -INFINITY; }
        } else {
          for (int a = 0; a < 3; ++a) { lo[a][k] = INFINITY; hi[a][k] = -INFINITY; }
          ref[k] = kRefEmpty;
        }
      }
      float4* d = &A.nodes[(size_t)w.dev * kNodeF4];
      for (int a = 0; a < 3; ++a) {
        d[a] = make_float4(lo[a][0], lo[a][1], lo[a][2], lo[a][3]);
        d[3 + a] = make_float4(hi[a][0], hi[a][1], hi[a][2], hi[a][3]);
      }
      d[6] = make_float4(as_float(ref[0]), as_float(ref[1]), as_float(ref[2]), as_float(ref[3]));
    }
  }
  // The first kTopNodes nodes in breadth-first order from the root come first in
  // the array (the traversal kernel keeps them in LDS); the others keep their
  // depth-first order.
  {
    const int n_nodes = (int)(A.nodes.size() / kNodeF4);
    std::vector<int> order;
    std::vector<char> taken((size_t)n_nodes, 0);
    if (n_nodes > 0) { order.push_back(0); taken[0] = 1; }
    for (size_t head = 0; head < order.size() && (int)order.size() < kTopNodes; ++head) {
      const float4 rf = A.nodes[(size_t)order[head] * kNodeF4 + 6];
      const float refs[4] = {rf.x, rf.y, rf.z, rf.w};
      for (int k = 0; k < 4 && (int)order.size() < kTopNodes; ++k) {
        int ref; memcpy(&ref, &refs[k], 4);
        if (ref >= 0 && !taken[ref]) { order.push_back(ref); taken[ref] = 1; }
      }
    }
    for (int i = 0; i < n_nodes; ++i)
      if (!taken[i]) order.push_back(i);
    std::vector<int> new_id((size_t)n_nodes);
    for (int i = 0; i < n_nodes; ++i) new_id[order[i]] = i;
    std::vector<float4> moved(A.nodes.size());
    for (int i = 0; i < n_nodes; ++i) {
      const float4* src = &A.nodes[(size_t)order[i] * kNodeF4];
      float4* dst = &moved[(size_t)i * kNodeF4];
      for (int k = 0; k < kNodeF4; ++k) dst[k] = src[k];
      float* refs = &dst[6].x;
      for (int k = 0; k < 4; ++k) {
        int ref; memcpy(&ref, &refs[k], 4);
        if (ref >= 0) { ref = new_id[ref]; memcpy(&refs[k], &ref, 4); }
      }
    }
    A.nodes.swap(moved);
  }
  for (size_t i = 0; i < np; ++i) {
    const Prim& p = sc.prims[i];
    float4* ir = &A.isect[i * kIsectF4];
    float4* sh = &A.shade[i * kShadeF4];
    if (p.type == GT_TRIANGLE) {
      f3 p1 = p.v[0].v, p2 = p.v[1].v, p3 = p.v[2].v;
      float maxE = fps::fmax_(fps::fmax_(maxabs(p1), maxabs(p2)), maxabs(p3));
      ir[0] = mk4(p1, maxE);
      ir[1] = mk4(p2 - p1, maxabs(p1));
      ir[2] = mk4(p3 - p1, as_float(tail[i]));
      sh[0] = mk4(p.v[0].n, p.v[0].uv.x);
      sh[1] = mk4(p.v[1].n, p.v[0].uv.y);
      sh[2] = mk4(p.v[2].n, p.v[1].uv.x);
      sh[3] = mk4(p.v[0].t, p.v[1].uv.y);
      sh[4] = mk4(p.v[1].t, p.v[2].uv.x);
      sh[5] = mk4(p.v[2].t, p.v[2].uv.y);
    } else if (p.type == GT_SPHERE) {
      ir[0] = mk4(p.p0, p.w0);
      ir[2] = make_float4(0, 0, 0, as_float(tail[i]));
      sh[0] = mk4(p.p0, p.w0);
    } else {
      ir[0] = mk4(p.p0, p.w0);
      ir[1] = mk4(p.p1, p.w1);
      ir[2] = make_float4(0, 0, 0, as_float(tail[i]));
      sh[0] = mk4(p.p0, p.w0);
      sh[1] = mk4(p.p1, p.w1);
    }
    sh[6] = make_float4(as_float(p.mat), as_float(p.light), as_float(p.med_in), as_float(p.med_out));
    // (the material's type rides along: the shade stage sorts hits by it before it loads anything else)
    const int mat_type = p.mat >= 0 && p.mat < (int)sc.materials.size() ? sc.materials[p.mat].type : -1;
    sh[7] = make_float4(as_float((uint32_t)p.type & kPrimTypeMask), as_float(mat_type), 0, 0);
  }
  A.lights.resize(sc.lights.size() * kLightF4);
  for (size_t i = 0; i < sc.lights.size(); ++i) {
    const Light& l = sc.lights[i];
    const Prim& p = sc.prims[l.prim];
    f3 e1 = p.v[1].v - p.v[0].v, e2 = p.v[2].v - p.v[0].v;
    // Triangle.h:81-85 as the device evaluates it (dot and cross fused, fpspec.h): the kernels read this value where the
    // reference's device code calls Area(); the light distribution, built on the host, keeps the unfused scene.h form
    f3 c = f3{fps::diffprod_(e1.y, e2.z, e1.z, e2.y), fps::diffprod_(e1.z, e2.x, e1.x, e2.z), fps::diffprod_(e1.x, e2.y, e1.y, e2.x)};
    float area = sqrtf(fps::dot3_(c.x, c.y, c.z, c.x, c.y, c.z)) * 0.5f;
    float4* d = &A.lights[i * kLightF4];
    d[0] = mk4(p.v[0].v, area);
    d[1] = mk4(p.v[1].v, l.radiance.x);
    d[2] = mk4(p.v[2].v, l.radiance.y);
    d[3] = mk4(p.v[0].n, l.radiance.z);
    d[4] = mk4(p.v[1].n, as_float(l.prim));
    d[5] = mk4(p.v[2].n, as_float(l.medium));
  }
  A.mats.resize(std::max<size_t>(1, sc.materials.size()) * kMatF4, make_float4(0, 0, 0, 0));
  for (size_t i = 0; i < sc.materials.size(); ++i) {
    const Material& m = sc.materials[i];
    float4* d = &A.mats[i * kMatF4];
    d[0] = make_float4(as_float(m.type), m.alphaU, m.alphaV, m.insideIOR);
    d[1] = make_float4(m.outsideIOR, as_float(m.tex), 0.f, 0.f);
    d[2] = mk4(m.k, 0.f);
    d[3] = mk4(m.eta, 0.f);
    d[4] = mk4(m.diffuse, 0.f);
    d[5] = mk4(m.specular, 0.f);
  }
  A.media.resize(std::max<size_t>(1, sc.media.size()) * kMediumF4, make_float4(0, 0, 0, 0));
  for (size_t i = 0; i < sc.media.size(); ++i) {
    const Medium& m = sc.media[i];
    float4* d = &A.media[i * kMediumF4];
    d[0] = make_float4(as_float(m.nx), as_float(m.ny), as_float(m.nz), m.g);
    d[1] = mk4(m.sigmaS, as_float(m.type));
    d[2] = mk4(m.sigmaT, m.invMaxDensity);
    d[3] = mk4(m.p0, as_float((int)A.density.size()));
    d[4] = mk4(m.p1, as_float(m.iterMax));
    A.density.insert(A.density.end(), m.density.begin(), m.density.end());
  }
  for (const Texture& t : sc.textures) {
    A.tex_info.push_back((int)A.tex_pool.size());
    A.tex_info.push_back(t.w);
    A.tex_info.push_back(t.h);
    size_t n = (size_t)t.w * t.h;
    for (size_t k = 0; k < n; ++k) {
      const uint8_t* c = &t.rgba[k * 4];
      A.tex_pool.push_back((uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16) | ((uint32_t)c[3] << 24));
    }
  }
  if (sc.infinite.valid) {
    A.env.resize(sc.infinite.data.size() * 3);
    for (size_t k = 0; k < sc.infinite.data.size(); ++k) {
      A.env[3 * k] = sc.infinite.data[k].x; A.env[3 * k + 1] = sc.infinite.data[k].y; A.env[3 * k + 2] = sc.infinite.data[k].z;
    }
  }
}

constexpr int kDefaultStreams = 3;
constexpr uint32_t kBdptJoinsPerSample = 32;  // join slots per sample of a launch of k_bdpt<true> (rounds that find no room are evaluated by k_bdpt<false>)
constexpr size_t kBdptQueueNodes = 0;       // binary BVH nodes above which the joins' shadow rays are queued (it pays on the Cornell box too: 88 -> 102 Msamples/s)    // joins a wavefront stages before it asks for slots
constexpr int kTrQueuePasses = 3;  // Tr() segments of the volumetric integrator's shadow rays done on queues
constexpr int kMaxStreams = 8;
// path state (80 B) + extension queues (2 x 52 B) + shadow (36 B) + probe (52 B) + three staging queues (3 x 36 B)
constexpr double kBytesPerSlot = 380.0;

static float gamma_n(int n) {  // common.h:111-114: double arithmetic, float result
  const double me = (double)1.192092896e-07F * 0.5;
  return (float)((n * me) / (1 - n * me));
}

}  // namespace caster

using namespace caster;

struct caster_scene {
  Scene scene;
};

namespace caster {
bool build_bvh_gpu(int device, const std::vector<BBox>& boxes, const BBox& root_box, std::vector<Node>& nodes,
                   std::vector<int>& order, std::string& err);  // bvh_build.hip
}

// One render stream with everything a batch of paths in flight needs.  Batches
// are dealt to the streams in turn, so the tail of one batch's kernels (few rays
// left in the late bounces, single-thread bookkeeping launches) overlaps the
// other stream's work; the scene arrays and the film are shared.
struct Lane {
  hipStream_t stream = nullptr;
  PathState P{};
  RayQueue ext[2]{}, shadow{}, probe{};
  RayQueue tr1{};  // volumetric integrator on queues: shadow rays going on past an index-matched boundary
  uint32_t* counters = nullptr;  // queue lengths: ext0, ext1, shadow, probe, tr1; then 3 x kWorkShards work counters
  WalkAux aux{};
  StageArea stage{};
  BdptStore bdpt{};  // sub-path vertices of the bidirectional integrator, one column per lane of its grid
  BdptJoins joins{}; // ... and its joins that wait for their shadow rays (bdpt_queues)
  uint32_t* join_n = nullptr;   // joins the present launch of k_bdpt handed slots to
  Stats* stats = nullptr;
  hipEvent_t finish_done = nullptr;  // recorded after the lane's latest k_finish
  // batch under way between the two halves of its submission
  uint32_t batch_first = 0, batch_n = 0;
  int batch_cur = 0, batch_b = 0;
};

struct caster_renderer {
  // A renderer made by caster_begin_multi owns one renderer per device and nothing else: every call on it is
  // dealt to them (each from a host thread of its own, with that device current) and their films meet on
  // the first one's device when the film is read.
  std::vector<caster_renderer*> parts;
  int shard_mode = CASTER_SHARD_TILE;
  float* group_sum = nullptr;    // on parts[0]'s device: the films of all parts added up
  float* group_peer = nullptr;   // on parts[0]'s device: one part's film on its way in
  std::string part_error;        // what the last failing part reported
  int device = 0;
  std::vector<Lane> lanes;
  uint32_t next_lane = 0;
  hipEvent_t last_finish = nullptr;  // the film is accumulated in iteration order: each k_finish waits for the one before
  std::vector<void*> allocs;
  DevScene S{};
  uint32_t work_init = 0;
  int walk_grid = 0;
  int shade_grid = 0;
  int bdpt_grid = 0;  // workgroups of k_bdpt (one sample per lane at a time)
  bool lt_queues = false;   // light tracing with its connection rays on queues (no density grid in the scene)
  bool bdpt_queues = false; // bidirectional integrator with its joins' shadow rays on queues (large scenes without a density grid)
  bool probe_any = false;   // path tracer: probe rays ride the shadow queue as occlusion rays (few emitters, no environment)
  bool vpt_queues = false;  // volumetric integrator with its shadow and probe rays on queues
  bool vpt_hetero = false;  // ... in a scene with a density grid: the vertex in three phases (volpath_hetero.h)
  float* acc = nullptr;
  float* film_tmp = nullptr;
  int32_t* ids = nullptr;
  void* static_scratch = nullptr;  // this device's copy of `scratch` (caster_trace_rays_static)
  int iters_in_flight = 1;
  uint32_t pool_slots = 0;
  uint32_t iterations = 0;
  uint64_t samples = 0;
  DevScene S_full{};         // S with the whole image as the row set (first-hit maps are always of the whole image)
  uint32_t live_pixels = 0;  // pixels on the scanlines this renderer traces
  int batch_iters = 1;       // iterations per batch: the path pool holds iters_in_flight iterations of the whole image,
                             // and more of a part of it
  int n_cu = 256;
  // Iterations traced ahead of the calls that ask for them one at a time (caster_set_lookahead): a batch of
  // `ahead_n` iterations from `ahead_first` sits finished-but-not-accumulated in lane `ahead_lane`'s pool, and
  // `ahead_done` of them have been added to the film.
  int lookahead = 0;
  int ahead_lane = -1;
  uint32_t ahead_first = 0, ahead_n = 0, ahead_done = 0;
  // optional per-kernel-class timing with HIP events on the render streams
  bool profiling = false;
  struct Timed { int cls; hipEvent_t a, b; };
  std::vector<Timed> pending_events;
  std::vector<hipEvent_t> free_events;
  double class_ms[CASTER_N_KERNEL_CLASSES] = {0};
  uint64_t class_launches[CASTER_N_KERNEL_CLASSES] = {0};
};

namespace caster {

template <class T>
static int dev_alloc(caster_renderer* r, T** out, size_t count) {
  void* p = nullptr;
  size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  {
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
      g_out_of_memory = e == hipErrorOutOfMemory;
      return fail(CASTER_ERR_DEVICE, "hipMalloc of " + std::to_string(bytes) + " bytes: " + hipGetErrorString(e));
    }
  }
  r->allocs.push_back(p);
  *out = (T*)p;
  return CASTER_OK;
}
template <class T>
static int dev_upload(caster_renderer* r, const T** out, const std::vector<T>& v) {
  T* p = nullptr;
  int rc = dev_alloc(r, &p, v.size());
  if (rc) return rc;
  if (!v.empty()) HIPCHECK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = p;
  return CASTER_OK;
}

#ifndef CASTER_PROBE_ANY_MAX_SLOTS
#define CASTER_PROBE_ANY_MAX_SLOTS 0xffffffffu  // path slots in flight up to which probe rays ride the shadow queue (every batch size)
#endif
static int alloc_queue(caster_renderer* r, RayQueue& q, uint32_t n, bool with_hit, uint32_t* counter) {
  int rc;
  if ((rc = dev_alloc(r, &q.o, n))) return rc;
  if ((rc = dev_alloc(r, &q.d, n))) return rc;
  if ((rc = dev_alloc(r, &q.slot, n))) return rc;
  q.hit = nullptr;
  if (with_hit && (rc = dev_alloc(r, &q.hit, n))) return rc;
  q.count = counter;
  return CASTER_OK;
}

static int alloc_lane(caster_renderer* r, Lane& ln) {
  int rc;
  // the bidirectional integrator keeps no paths in queues: its queues only serve the first-hit map
  const uint32_t n = r->S.integrator == IT_BDPT ? (uint32_t)r->S.slots_per_iter : r->pool_slots;
  HIPCHECK(hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
  HIPCHECK(hipEventCreateWithFlags(&ln.finish_done, hipEventDisableTiming));
  if ((rc = dev_alloc(r, &ln.P.beta_rng, n))) return rc;
  if ((rc = dev_alloc(r, &ln.P.li_flags, n))) return rc;
  if ((rc = dev_alloc(r, &ln.P.pend_x, n))) return rc;
  if ((rc = dev_alloc(r, &ln.P.pend_y, n))) return rc;
  if ((rc = dev_alloc(r, &ln.P.beta_prev, n))) return rc;
  if ((rc = dev_alloc(r, &ln.counters, 64 + 3 * kWorkShards * kWorkPad))) return rc;
  HIPCHECK(hipMemset(ln.counters, 0, (64 + 3 * kWorkShards * kWorkPad) * sizeof(uint32_t)));
  if ((rc = alloc_queue(r, ln.ext[0], n, true, ln.counters + 0))) return rc;
  if ((rc = alloc_queue(r, ln.ext[1], n, true, ln.counters + 1))) return rc;
  // (with the probe rays among the shadow rays a path can append two entries to that queue, none to the other)
  // (bidirectional integrator with queued joins: the shadow queues, the transmittance records and the staging regions
  // of k_tr_step hold join slots, kBdptJoinsPerSample for every sample of a launch)
  uint32_t joins_per_sample = kBdptJoinsPerSample;
  if (const char* e = getenv("CASTER_BDPT_JOIN_SLOTS")) {   // (a knob for tests: few slots send rounds to the in-place kernel)
    int v = atoi(e);
    if (v >= 1 && v <= 64) joins_per_sample = (uint32_t)v;
  }
  if (r->bdpt_queues) {
    // (a join slot costs some 170 bytes over the queues, records and staging regions; a lane takes an eighth of the free
    // memory for them at most -- the staircase at 1080p: 33 Msamples/s in place, 37 with 16 slots per sample, 53 with 32 or 64)
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
      while (joins_per_sample > 4 && (size_t)joins_per_sample * n * 170 > free_b / 8) joins_per_sample /= 2;
  }
  const uint32_t nq = r->bdpt_queues ? joins_per_sample * n : n;
  if ((rc = alloc_queue(r, ln.shadow, r->probe_any ? 2 * n : nq, r->vpt_queues || r->lt_queues || r->bdpt_queues, ln.counters + 2))) return rc;
  if ((rc = alloc_queue(r, ln.probe, r->probe_any ? kWave : n, true, ln.counters + 3))) return rc;
  if (r->vpt_queues || r->lt_queues) {
    if ((rc = alloc_queue(r, ln.tr1, n, true, ln.counters + 4))) return rc;
    if ((rc = dev_alloc(r, &ln.P.pend_z, n))) return rc;
    if ((rc = dev_alloc(r, &ln.P.pend_w, n))) return rc;
    if ((rc = dev_alloc(r, &ln.P.pend_v, n))) return rc;
  }
  if (r->bdpt_queues) {
    if ((rc = alloc_queue(r, ln.tr1, nq, true, ln.counters + 4))) return rc;
    if ((rc = dev_alloc(r, &ln.P.pend_w, nq))) return rc;
    ln.join_n = ln.counters + 5;
    ln.joins.q = ln.shadow;
    ln.joins.cap = nq;
    ln.joins.unused = ln.counters + 6;
    ln.joins.n_deferred = ln.counters + 7;
    if ((rc = dev_alloc(r, &ln.joins.val, nq))) return rc;
    if ((rc = dev_alloc(r, &ln.joins.deferred, (n + kWave - 1) / kWave))) return rc;   // (a launch covers an iteration: n samples)
  }
  // every shade wavefront owns a staging region large enough for all paths it can be given
  uint32_t rounds = (nq + kWave - 1) / kWave;
  uint32_t per = (rounds + (uint32_t)r->shade_grid - 1) / (uint32_t)r->shade_grid;
  ln.stage.region_cap = (per + CASTER_SHADE_CHUNK) * kWave;   // (chunks dealt round-robin: up to one chunk more than an even share)
  ln.stage.shadow_region_cap = r->probe_any ? 2 * ln.stage.region_cap : ln.stage.region_cap;
  uint32_t cap = ln.stage.region_cap * (uint32_t)r->shade_grid;
  if ((rc = alloc_queue(r, ln.stage.next, r->bdpt_queues ? kWave : cap, false, nullptr))) return rc;   // (only k_tr_step stages anything there)
  if ((rc = alloc_queue(r, ln.stage.shadow, ln.stage.shadow_region_cap * (uint32_t)r->shade_grid, false, nullptr))) return rc;
  if ((rc = alloc_queue(r, ln.stage.probe, r->probe_any || r->bdpt_queues ? kWave : cap, false, nullptr))) return rc;
  // stack spill columns, one per lane of the widest grid that walks the tree (the traversal kernel, and the shade-grid
  // and bidirectional kernels that walk in place); the traversal kernel keeps kTraceRays stacks per lane
  ln.aux.spill_stride = (uint32_t)std::max(r->walk_grid, std::max(r->shade_grid, r->bdpt_grid)) * kWave;
  if ((rc = dev_alloc(r, &ln.aux.spill, (size_t)ln.aux.spill_stride * kSpillDepth * kTraceRays))) return rc;
  ln.aux.work = ln.counters + 64;
  if (r->S.integrator == IT_BDPT) {
    ln.bdpt.stride = (uint32_t)r->bdpt_grid * kWave;
    if ((rc = dev_alloc(r, &ln.bdpt.v, (size_t)kBdptVertF4 * 2 * kBdptPathVerts * ln.bdpt.stride))) return rc;
  }
  if ((rc = dev_alloc(r, &ln.stats, 1))) return rc;
  HIPCHECK(hipMemset(ln.stats, 0, sizeof(Stats)));
  return CASTER_OK;
}

static int begin_impl(const Scene& sc, int device_id, int iters_in_flight, int n_streams, caster_renderer* r) {
  r->device = device_id;
  HIPCHECK(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  HIPCHECK(hipGetDeviceProperties(&prop, device_id));
  r->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  {
    int rc_s = static_scratch_of(device_id, &r->static_scratch);
    if (rc_s) return rc_s;
  }

  HostArrays A;
  build_arrays(sc, A);
  DevScene& S = r->S;
  int rc;
  if ((rc = dev_upload(r, &S.nodes, A.nodes))) return rc;
  if ((rc = dev_upload(r, &S.isect, A.isect))) return rc;
  if ((rc = dev_upload(r, &S.shade, A.shade))) return rc;
  if ((rc = dev_upload(r, &S.lights, A.lights))) return rc;
  if ((rc = dev_upload(r, &S.mats, A.mats))) return rc;
  if ((rc = dev_upload(r, &S.media, A.media))) return rc;
  if ((rc = dev_upload(r, &S.density, A.density))) return rc;
  S.has_hetero = sc.has_heterogeneous ? 1 : 0;
  if ((rc = dev_upload(r, &S.light_cdf, sc.light_cdf))) return rc;
  if ((rc = dev_upload(r, &S.tex_pool, A.tex_pool))) return rc;
  if ((rc = dev_upload(r, &S.tex_info, A.tex_info))) return rc;
  if ((rc = dev_upload(r, &S.inf.data, A.env))) return rc;
  S.n_prims = (int)sc.prims.size();
  // (the traversal kernel addresses node records with 32-bit byte offsets from the array's start)
  if (A.nodes.size() * sizeof(float4) > (size_t)UINT32_MAX)
    return fail(CASTER_ERR_ARG, "caster_begin: the tree's node array exceeds 4 GiB (" + std::to_string(A.nodes.size() / kNodeF4) + " wide nodes)");
  S.n_nodes = (int)(A.nodes.size() / kNodeF4);
  S.n_lights = (int)sc.lights.size();
  S.n_cdf = (int)sc.light_cdf.size();
  const Camera& c = sc.camera;
  DevCamera& dc = S.cam;
  const f3* src[4] = {&c.position, &c.u, &c.v, &c.w};
  float* dst[4] = {dc.pos, dc.u, dc.v, dc.w};
  for (int k = 0; k < 4; ++k) { dst[k][0] = src[k]->x; dst[k][1] = src[k]->y; dst[k][2] = src[k]->z; }
  dc.res_x = c.res_x; dc.res_y = c.res_y; dc.distance = c.distance;
  dc.apertureRadius = c.apertureRadius; dc.focalDistance = c.focalDistance;
  dc.half_w = c.half_w; dc.half_h = c.half_h; dc.p2s_x = c.p2s_x; dc.p2s_y = c.p2s_y; dc.ratio = c.ratio; dc.area = c.area;
  dc.environment = c.environment; dc.medium = c.medium; dc.filmic = c.filmic;
  S.inf.valid = sc.infinite.valid; S.inf.w = sc.infinite.w; S.inf.h = sc.infinite.h;
  S.inf.radius = sc.infinite.radius;
  const f3* isrc[3] = {&sc.infinite.u, &sc.infinite.v, &sc.infinite.w_};
  float* idst[3] = {S.inf.u, S.inf.v, S.inf.w_};
  for (int k = 0; k < 3; ++k) { idst[k][0] = isrc[k]->x; idst[k][1] = isrc[k]->y; idst[k][2] = isrc[k]->z; }
  S.epsilon = sc.epsilon;
  S.g1 = gamma_n(1); S.g2 = gamma_n(2); S.g3 = gamma_n(3);
  S.box_scale = 1 + 2 * S.g3;
  S.far_t = sc.far_t;
  {
    // Bounds of the emitters, for rejecting BSDF-sampled probe rays that cannot reach one.  The
    // pad (1e-3 of the box's size and position) is some four orders of magnitude above the
    // rounding error of the ray/triangle and ray/box tests, so a ray that any of those tests could
    // report as hitting an emitter passes this box.
    BBox eb;
    for (const Light& l : sc.lights)
      for (int k = 0; k < 3; ++k) eb.expand(sc.prims[l.prim].v[k].v);
    f3 d = eb.hi - eb.lo;
    float pad = sc.lights.empty() ? 0.f
                                  : 1e-3f * (fps::fmax_(fps::fmax_(d.x, d.y), d.z) + fps::fmax_(maxabs(eb.lo), maxabs(eb.hi))) + 1e-6f;
    S.emit_lo[0] = eb.lo.x - pad; S.emit_lo[1] = eb.lo.y - pad; S.emit_lo[2] = eb.lo.z - pad;
    S.emit_hi[0] = eb.hi.x + pad; S.emit_hi[1] = eb.hi.y + pad; S.emit_hi[2] = eb.hi.z + pad;
  }
  S.integrator = sc.integrator; S.maxDepth = sc.maxDepth; S.maxDist = sc.maxDist;
  S.width = sc.width; S.height = sc.height;
  S.tiles_x = (sc.width + 7) / 8; S.tiles_y = (sc.height + 7) / 8;
  S.slots_per_iter = S.tiles_x * S.tiles_y * 64;
  S.row_y0 = 0; S.row_y1 = sc.height; S.row_rows = 1; S.row_stride = 1;
  S.tile_row0 = 0; S.tile_rows = 1; S.tile_stride = 1;
  r->live_pixels = (uint32_t)sc.width * (uint32_t)sc.height;

  const bool default_pool = iters_in_flight <= 0;
  if (default_pool) {
    // Late bounces keep few rays alive: a batch needs tens of millions of paths for their launches to fill the
    // chip.  The default is 64 M path slots per stream (32 iterations of 1080p; 76 GB with three streams), which
    // costs 1 % on the 1.37 M-triangle benchmark scene against twice that and nothing on the smaller scenes; a
    // caller that wants the last percent states iters_in_flight (128 M slots x 380 B per stream = 150 GB; 192 M
    // measured no faster), one that shares the card states less.
    long long target = 64ll << 20;
    iters_in_flight = (int)std::max<long long>(1, std::min<long long>(64, target / S.slots_per_iter));
  }
  if (n_streams <= 0) n_streams = kDefaultStreams;
  if (n_streams > kMaxStreams) n_streams = kMaxStreams;
  if (default_pool) {
    // the default pools take at most a quarter of the memory that is free right now (a host that links this
    // library may hold something of its own, and other renderers or processes may share the GPU); a caller
    // that states iters_in_flight gets what it asked for or an allocation error
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > 0) {
      const double per_iter = (double)S.slots_per_iter * kBytesPerSlot * n_streams;
      long long fit = (long long)((double)free_b * 0.25 / per_iter);
      iters_in_flight = (int)std::max<long long>(1, std::min<long long>(iters_in_flight, fit));
    }
  }
  // queue entries and path slots are 32-bit signed indices (the shadow queue holds two per slot when the probe
  // rays ride it): a batch beyond that is refused, not wrapped
  if ((uint64_t)S.slots_per_iter * (uint64_t)iters_in_flight * 2u > (uint64_t)INT32_MAX)
    return fail(CASTER_ERR_ARG, "caster_begin: iters_in_flight * pixels exceeds the 2^30 path slots a renderer can index");
  r->iters_in_flight = iters_in_flight;
  r->batch_iters = iters_in_flight;
  r->pool_slots = (uint32_t)S.slots_per_iter * (uint32_t)iters_in_flight;
  // shade grid: one wavefront per block, as many as are resident at once
  {
    int per_cu = 8;
    // (CASTER_VPT_INKERNEL=1 in the environment keeps the one-kernel vertex for scenes with a density grid: a knob
    // for tests and measurements)
    const char* in_kernel = getenv("CASTER_VPT_INKERNEL");
    r->vpt_queues = sc.integrator == IT_VPT && !(sc.has_heterogeneous && in_kernel && in_kernel[0] == '1');
    r->vpt_hetero = r->vpt_queues && sc.has_heterogeneous;
    // Every probe ray of the path tracer then starts out with the nearest emitter on its line, and can be
    // settled in k_shade_pt<true> but for its visibility (see there): two launches less per bounce, for a
    // heavier shade kernel.  Against a closest-hit probe queue with a resolve kernel of its own, 1080p on
    // 1.37 M triangles: +19 % with one iteration in flight, +7 % with four, +3 % with sixteen, and with
    // sixty-four +1 ... +3 % (1757 against 1696-1742 Msamples/s; 1.5 % slower before the shade stage
    // shaded hits only): every batch size takes it.
    // (CASTER_PROBE_QUEUE=1 in the environment keeps the probe queue: a knob for tests and measurements)
    const char* keep_queue = getenv("CASTER_PROBE_QUEUE");
    r->probe_any = !(keep_queue && keep_queue[0] == '1') && CASTER_PROBE_CULL && sc.integrator == IT_PT && !sc.infinite.valid &&
                   (int)sc.lights.size() <= kProbeCullLights && (uint64_t)r->pool_slots <= (uint64_t)CASTER_PROBE_ANY_MAX_SLOTS;
    // light tracing sends its connection rays through the traversal kernel unless a medium is a
    // density grid (its transmittance draws random numbers) -- or the scene is a handful of
    // primitives without media, where walking a three-level tree in place beats three more
    // launches per round (Cornell box: 513 against 443 Msamples/s; 1.37 M triangles: 420 against 775)
    r->lt_queues = sc.integrator == IT_LT && !sc.has_heterogeneous && (sc.nodes.size() > 128 || !sc.media.empty());
    // the bidirectional integrator sends the shadow rays of its joins through the traversal kernel in scenes
    // without a density grid (see BdptJoins; CASTER_BDPT_INKERNEL=1 in the environment keeps them in k_bdpt: a knob
    // for tests and measurements)
    const char* bdpt_in_kernel = getenv("CASTER_BDPT_INKERNEL");
    size_t queue_nodes = kBdptQueueNodes;
    if (const char* e = getenv("CASTER_BDPT_QUEUE_NODES")) queue_nodes = (size_t)atol(e);   // (measurement knob)
    r->bdpt_queues = sc.integrator == IT_BDPT && !sc.has_heterogeneous && sc.nodes.size() > queue_nodes &&
                     !(bdpt_in_kernel && bdpt_in_kernel[0] == '1');
    const void* fn = sc.integrator == IT_AO ? (const void*)k_shade_ao
                     : sc.integrator == IT_VPT ? (r->vpt_hetero ? (const void*)k_vpth<0> : r->vpt_queues ? (const void*)k_shade_vptq : (const void*)k_shade_vpt)
                     : sc.integrator == IT_LT ? (r->lt_queues ? (const void*)k_shade_lt<true> : (const void*)k_shade_lt<false>)
                                              : (const void*)k_shade_pt<false>;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kWave, 0) != hipSuccess || per_cu < 1) per_cu = 8;
    if (per_cu > 20) per_cu = 20;
    r->shade_grid = r->n_cu * per_cu;
  }
  // persistent traversal grid: as many wavefronts as LDS stacks and registers allow per CU
  {
    // per workgroup: kTraceBlockWaves stacks and one copy of the top of the tree
    int lds_per_block = kTraceBlockWaves * kTraceRays * kLdsDepth * kWave * 8 + std::max(kTopNodes, 1) * kTopF4 * 16;
    int by_lds = (160 * 1024) / lds_per_block * kTraceBlockWaves;
    int by_regs = CASTER_TRACE_WAVES * 4 / kTraceBlockWaves * kTraceBlockWaves;
    r->walk_grid = r->n_cu * std::max(kTraceBlockWaves, std::min(by_lds, by_regs));
    // tuning knobs: fewer resident wavefronts per CU for the traversal / the shade kernels (never more than fit)
    if (const char* e = getenv("CASTER_WALK_PER_CU")) {
      int v = atoi(e) / kTraceBlockWaves * kTraceBlockWaves;
      if (v >= kTraceBlockWaves && r->n_cu * v < r->walk_grid) r->walk_grid = r->n_cu * v;
    }
    if (const char* e = getenv("CASTER_SHADE_PER_CU")) {
      int v = atoi(e);
      if (v >= 1 && r->n_cu * v < r->shade_grid) r->shade_grid = r->n_cu * v;
    }
  }
  r->work_init = 0;  // the work counters count from the end of k_trace's static assignment, whose size it derives from the queue length
  if (sc.integrator == IT_BDPT) {
    // one sample per lane at a time: as many wavefronts as are resident at once, within the
    // traversal grid (whose stack spill columns k_bdpt's lanes use)
    int per_cu = 4;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, r->bdpt_queues ? (const void*)k_bdpt<true> : (const void*)k_bdpt<false>, kWave, 0) != hipSuccess || per_cu < 1) per_cu = 4;
    r->bdpt_grid = std::min(r->n_cu * std::min(per_cu, 24), r->walk_grid);
  }
  r->lanes.resize((size_t)n_streams);
  for (Lane& ln : r->lanes)
    if ((rc = alloc_lane(r, ln))) return rc;
  size_t npix = (size_t)sc.width * sc.height;
  if ((rc = dev_alloc(r, &r->acc, npix * 3))) return rc;
  HIPCHECK(hipMemset(r->acc, 0, npix * 3 * sizeof(float)));
  if ((rc = dev_alloc(r, &r->film_tmp, npix * 3))) return rc;
  if ((rc = dev_alloc(r, &r->ids, npix))) return rc;
  r->S_full = r->S;
  return CASTER_OK;
}

static inline LaunchCtx ctx_of(caster_renderer* r, const Lane& ln) { return LaunchCtx{r->S, ln.P, ln.stats}; }

static int sync_lanes(caster_renderer* r) {
  for (Lane& ln : r->lanes) HIPCHECK(hipStreamSynchronize(ln.stream));
  return CASTER_OK;
}

static hipEvent_t take_event(caster_renderer* r) {
  if (!r->free_events.empty()) { hipEvent_t e = r->free_events.back(); r->free_events.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
// Brackets one kernel launch with events when profiling is on.
struct ScopedTimer {
  caster_renderer* r; hipStream_t st; int cls; hipEvent_t a = nullptr, b = nullptr;
  ScopedTimer(caster_renderer* r_, hipStream_t st_, int cls_) : r(r_), st(st_), cls(cls_) {
    if (r->profiling) { a = take_event(r); b = take_event(r); (void)hipEventRecord(a, st); }
  }
  ~ScopedTimer() {
    if (r->profiling) { (void)hipEventRecord(b, st); r->pending_events.push_back({cls, a, b}); }
  }
};
static void collect_timers(caster_renderer* r) {
  for (auto& t : r->pending_events) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) { r->class_ms[t.cls] += ms; r->class_launches[t.cls]++; }
    r->free_events.push_back(t.a);
    r->free_events.push_back(t.b);
  }
  r->pending_events.clear();
}

// Enqueues one batch of `n` iterations starting at `first` on stream `lane`, in two halves:
// `tail` false submits the camera rays and the first maxDepth rounds, which need nothing from
// the host; `tail` true (same lane, after the heads of the other lanes have been submitted, so
// that they run meanwhile) submits what does -- the volumetric integrator's extra rounds, whose
// number depends on a queue length -- and the film accumulation.
static int finish_iterations(caster_renderer* r, uint32_t lane, uint32_t n, uint32_t k0, uint32_t k1);
static int enqueue_batch(caster_renderer* r, uint32_t lane, bool tail, uint32_t first, uint32_t n, bool finish = true) {
  const DevScene& S = r->S;
  Lane& ln = r->lanes[lane];
  if (tail) { first = ln.batch_first; n = ln.batch_n; }
  hipStream_t st = ln.stream;
  LaunchCtx L = ctx_of(r, ln);
  const uint32_t total = (uint32_t)S.slots_per_iter * n;
  const int walk_grid = r->walk_grid;
  auto aux_for = [&](int which) { WalkAux a = ln.aux; a.work = ln.counters + 64 + which * kWorkShards * kWorkPad; return a; };
  const int shade_grid = r->shade_grid;
  const int gen_grid = (int)std::min<uint32_t>((total + kWave - 1) / kWave, (uint32_t)r->n_cu * 32);
  int cur = tail ? ln.batch_cur : 0;
  if (S.integrator == IT_BDPT) {
    // the whole sample runs in one kernel and splats into the film with float atomics (from every
    // stream at once, as light tracing does): nothing to submit in the second half
    if (!tail && !r->bdpt_queues) {
      ScopedTimer t(r, st, CASTER_K_SHADE);
      hipLaunchKernelGGL(k_bdpt<false>, dim3(r->bdpt_grid), dim3(kWave), 0, st, L, first, n, 0u, total, r->acc, ln.aux, ln.bdpt, BdptJoins{},
                         (const uint32_t*)nullptr, (const uint32_t*)nullptr);
      ln.batch_first = first; ln.batch_n = n;
    }
    if (!tail && r->bdpt_queues) {
      // an iteration's worth of samples per launch (the join slots are sized for that), then the joins' shadow rays a
      // Tr() segment per pass of the traversal kernel, then the splats
      const uint32_t per_launch = (uint32_t)S.slots_per_iter;
      for (uint32_t begin = 0; begin < total; begin += per_launch) {
        const uint32_t end = begin + per_launch < total ? begin + per_launch : total;
        {
          ScopedTimer t(r, st, CASTER_K_SHADE);
          hipLaunchKernelGGL(k_bdpt<true>, dim3(r->bdpt_grid), dim3(kWave), 0, st, L, first, n, begin, end, r->acc, ln.aux, ln.bdpt, ln.joins,
                             (const uint32_t*)nullptr, (const uint32_t*)nullptr);
        }
        hipLaunchKernelGGL(k_join_count, dim3(1), dim3(1), 0, st, ln.shadow.count, ln.joins.cap, ln.join_n, ln.joins.unused, ln.stats);
        {
          // the rounds that found the queue too full (none, as a rule), with their shadow rays walked in place
          ScopedTimer t(r, st, CASTER_K_SHADE);
          hipLaunchKernelGGL(k_bdpt<false>, dim3(r->bdpt_grid), dim3(kWave), 0, st, L, first, n, 0u, 0u, r->acc, ln.aux, ln.bdpt, BdptJoins{},
                             (const uint32_t*)ln.joins.deferred, (const uint32_t*)ln.joins.n_deferred);
        }
        hipLaunchKernelGGL(k_set_count, dim3(1), dim3(1), 0, st, ln.joins.n_deferred, 0u);
        RayQueue* tq[2] = {&ln.shadow, &ln.tr1};
        int tcur = 0;
        for (int it = 0; it < kTrQueuePasses; ++it) {
          hipLaunchKernelGGL(k_add_stat, dim3(1), dim3(1), 0, st, ln.stats, 3, (const uint32_t*)tq[tcur]->count);
          hipLaunchKernelGGL(k_roll, dim3(1), dim3(1), 0, st, ln.stats, (const uint32_t*)nullptr, (const uint32_t*)nullptr,
                             (const uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u,
                             ln.counters + 64 + 1 * kWorkShards * kWorkPad, (uint32_t)kWorkShards, r->work_init, 0u);
          {
            ScopedTimer t(r, st, CASTER_K_SHADOW);
            hipLaunchKernelGGL(k_trace<false>, dim3(walk_grid / kTraceBlockWaves), dim3(kWave * kTraceBlockWaves), 0, st, L, *tq[tcur], aux_for(1));
          }
          hipLaunchKernelGGL(k_tr_step<false>, dim3(shade_grid), dim3(kWave), 0, st, L, *tq[tcur], *tq[tcur ^ 1], ln.stage);
          hipLaunchKernelGGL(k_set_count, dim3(1), dim3(1), 0, st, tq[tcur]->count, 0u);
          tcur ^= 1;
        }
        hipLaunchKernelGGL(k_tr_finish<false>, dim3(shade_grid), dim3(kWave), 0, st, L, *tq[tcur], ln.aux);
        hipLaunchKernelGGL(k_set_count, dim3(1), dim3(1), 0, st, tq[tcur]->count, 0u);
        {
          ScopedTimer t(r, st, CASTER_K_PROBE);
          hipLaunchKernelGGL(k_bdpt_resolve, dim3(r->n_cu * 8), dim3(256), 0, st, L, ln.joins, (const uint32_t*)ln.join_n, r->acc);
        }
      }
      ln.batch_first = first; ln.batch_n = n;
    }
    HIPCHECK(hipGetLastError());
    return CASTER_OK;
  }
  if (!tail) {
    {
      ScopedTimer t(r, st, CASTER_K_GENERATE);
      if (S.integrator == IT_LT)  // walks the first camera connection: one stack per wavefront of the shade grid
        if (r->lt_queues)
          hipLaunchKernelGGL(k_generate_lt<true>, dim3(r->shade_grid), dim3(kWave), 0, st, L, ln.ext[0], first, n,
                             r->acc, ln.aux, LtConn{ln.shadow, ln.probe, ln.stage});
        else
          hipLaunchKernelGGL(k_generate_lt<false>, dim3(r->shade_grid), dim3(kWave), 0, st, L, ln.ext[0], first, n,
                             r->acc, ln.aux, LtConn{});
      else
        hipLaunchKernelGGL(k_generate, dim3(gen_grid), dim3(kWave), 0, st, L, ln.ext[0], first, n);
    }
    hipLaunchKernelGGL(k_set_count, dim3(1), dim3(1), 0, st, ln.ext[0].count, total);
    // (light tracing on queues: the shadow queue and the list already hold the connections of the
    // paths' first vertices; both are empty at the end of every round)
    hipLaunchKernelGGL(k_roll, dim3(1), dim3(1), 0, st, ln.stats, (const uint32_t*)nullptr, (const uint32_t*)nullptr,
                       (const uint32_t*)nullptr, ln.ext[1].count, r->lt_queues ? (uint32_t*)nullptr : ln.shadow.count,
                       r->lt_queues ? (uint32_t*)nullptr : ln.probe.count, 0u, ln.counters + 64, 3u * kWorkShards, r->work_init, 0u);
  }
  const uint32_t live_rays = r->live_pixels * n;
  // One round of the volumetric integrator with everything on queues: the shadow rays are
  // advanced a Tr() segment at a time (closest-hit pass, then k_tr_step) until none is left.
  auto bounce_vptq = [&](int b) {
    RayQueue& q = ln.ext[cur];
    ShadeQueues out{ln.ext[cur ^ 1], ln.shadow, ln.probe};
    {
      ScopedTimer t(r, st, CASTER_K_EXTEND);
      hipLaunchKernelGGL(k_trace<false>, dim3(walk_grid / kTraceBlockWaves), dim3(kWave * kTraceBlockWaves), 0, st, L, q, aux_for(0));
    }
    const bool het = r->vpt_hetero;
    {
      ScopedTimer t(r, st, CASTER_K_SHADE);
      if (het) {
        // the free flights first, in a kernel whose lanes take their next path as soon as their tracking loop ends
        hipLaunchKernelGGL(k_vpth_flight, dim3(shade_grid * kTrackGridMul), dim3(kWave), 0, st, L, q);
        hipLaunchKernelGGL(k_vpth<0>, dim3(shade_grid), dim3(kWave), 0, st, L, q, ln.probe, out, ln.stage);
      } else hipLaunchKernelGGL(k_shade_vptq, dim3(shade_grid), dim3(kWave), 0, st, L, q, out, ln.stage);
    }
    RayQueue* tq[2] = {&ln.shadow, &ln.tr1};
    int tcur = 0;
    for (int it = 0; it < kTrQueuePasses; ++it) {
      hipLaunchKernelGGL(k_add_stat, dim3(1), dim3(1), 0, st, ln.stats, 3, (const uint32_t*)tq[tcur]->count);
      hipLaunchKernelGGL(k_roll, dim3(1), dim3(1), 0, st, ln.stats, (const uint32_t*)nullptr, (const uint32_t*)nullptr,
                         (const uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u,
                         ln.counters + 64 + 1 * kWorkShards * kWorkPad, (uint32_t)kWorkShards, r->work_init, 0u);
      {
        ScopedTimer t(r, st, CASTER_K_SHADOW);
        hipLaunchKernelGGL(k_trace<false>, dim3(walk_grid / kTraceBlockWaves), dim3(kWave * kTraceBlockWaves), 0, st, L, *tq[tcur], aux_for(1));
      }
      if (het) hipLaunchKernelGGL(k_tr_step_grid, dim3(shade_grid * kTrackGridMul), dim3(kWave), 0, st, L, *tq[tcur], *tq[tcur ^ 1], ln.stage);
      else hipLaunchKernelGGL(k_tr_step<false>, dim3(shade_grid), dim3(kWave), 0, st, L, *tq[tcur], *tq[tcur ^ 1], ln.stage);
      hipLaunchKernelGGL(k_set_count, dim3(1), dim3(1), 0, st, tq[tcur]->count, 0u);
      tcur ^= 1;
    }
    // whatever is still under way is walked to its end lane by lane
    if (het) hipLaunchKernelGGL(k_tr_finish<true>, dim3(shade_grid), dim3(kWave), 0, st, L, *tq[tcur], ln.aux);
    else hipLaunchKernelGGL(k_tr_finish<false>, dim3(shade_grid), dim3(kWave), 0, st, L, *tq[tcur], ln.aux);
    if (het) {
      // the paths that waited for their Tr(): a scattering event's new direction, a surface vertex's probe
      ScopedTimer t(r, st, CASTER_K_SHADE);
      hipLaunchKernelGGL(k_vpth<1>, dim3(shade_grid), dim3(kWave), 0, st, L, q, ln.probe, out, ln.stage);
    }
    {
      ScopedTimer t(r, st, CASTER_K_PROBE);
      hipLaunchKernelGGL(k_trace<false>, dim3(walk_grid / kTraceBlockWaves), dim3(kWave * kTraceBlockWaves), 0, st, L, ln.probe, aux_for(2));
    }
    if (het) {
      ScopedTimer t(r, st, CASTER_K_SHADE);
      hipLaunchKernelGGL(k_vpth<2>, dim3(shade_grid), dim3(kWave), 0, st, L, q, ln.probe, out, ln.stage);
    } else {
      hipLaunchKernelGGL(k_probe_resolve_vpt, dim3(shade_grid), dim3(kWave), 0, st, L, ln.probe);
    }
    hipLaunchKernelGGL(k_roll, dim3(1), dim3(1), 0, st, ln.stats, (const uint32_t*)q.count, (const uint32_t*)nullptr,
                       (const uint32_t*)ln.probe.count, q.count, tq[tcur]->count, ln.probe.count, b == 0 ? live_rays : 0u,
                       ln.counters + 64, 3u * kWorkShards, r->work_init, 0u);
    cur ^= 1;
  };
  // Light tracing on queues: the connection rays of one round, advanced a Tr() segment per
  // closest-hit pass as the volumetric integrator's shadow rays are, then splatted.
  auto lt_connections = [&]() {
    RayQueue* tq[2] = {&ln.shadow, &ln.tr1};
    int tcur = 0;
    for (int it = 0; it < kTrQueuePasses; ++it) {
      hipLaunchKernelGGL(k_add_stat, dim3(1), dim3(1), 0, st, ln.stats, 3, (const uint32_t*)tq[tcur]->count);
      hipLaunchKernelGGL(k_roll, dim3(1), dim3(1), 0, st, ln.stats, (const uint32_t*)nullptr, (const uint32_t*)nullptr,
                         (const uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u,
                         ln.counters + 64 + 1 * kWorkShards * kWorkPad, (uint32_t)kWorkShards, r->work_init, 0u);
      {
        ScopedTimer t(r, st, CASTER_K_SHADOW);
        hipLaunchKernelGGL(k_trace<false>, dim3(walk_grid / kTraceBlockWaves), dim3(kWave * kTraceBlockWaves), 0, st, L, *tq[tcur], aux_for(1));
      }
      hipLaunchKernelGGL(k_tr_step<false>, dim3(shade_grid), dim3(kWave), 0, st, L, *tq[tcur], *tq[tcur ^ 1], ln.stage);
      hipLaunchKernelGGL(k_set_count, dim3(1), dim3(1), 0, st, tq[tcur]->count, 0u);
      tcur ^= 1;
    }
    hipLaunchKernelGGL(k_tr_finish<false>, dim3(shade_grid), dim3(kWave), 0, st, L, *tq[tcur], ln.aux);
    hipLaunchKernelGGL(k_set_count, dim3(1), dim3(1), 0, st, tq[tcur]->count, 0u);
    {
      ScopedTimer t(r, st, CASTER_K_PROBE);
      hipLaunchKernelGGL(k_lt_resolve, dim3(shade_grid), dim3(kWave), 0, st, L, ln.probe, r->acc);
    }
    hipLaunchKernelGGL(k_set_count, dim3(1), dim3(1), 0, st, ln.probe.count, 0u);
  };
  if (!tail && r->lt_queues) lt_connections();
  auto bounce = [&](int b) {
    if (r->vpt_queues) { bounce_vptq(b); return; }
    RayQueue& q = ln.ext[cur];
    ShadeQueues out{ln.ext[cur ^ 1], ln.shadow, ln.probe};
    {
      ScopedTimer t(r, st, CASTER_K_EXTEND);
      hipLaunchKernelGGL(k_trace<false>, dim3(walk_grid / kTraceBlockWaves), dim3(kWave * kTraceBlockWaves), 0, st, L, q, aux_for(0));
    }
    {
      ScopedTimer t(r, st, CASTER_K_SHADE);
      if (S.integrator == IT_AO) hipLaunchKernelGGL(k_shade_ao, dim3(shade_grid), dim3(kWave), 0, st, L, q, out, ln.stage);
      else if (S.integrator == IT_VPT)
        hipLaunchKernelGGL(k_shade_vpt, dim3(shade_grid), dim3(kWave), 0, st, L, q, out, ln.stage, ln.aux);
      else if (S.integrator == IT_LT)
        if (r->lt_queues) hipLaunchKernelGGL(k_shade_lt<true>, dim3(shade_grid), dim3(kWave), 0, st, L, q, out, ln.stage, ln.aux, r->acc);
        else hipLaunchKernelGGL(k_shade_lt<false>, dim3(shade_grid), dim3(kWave), 0, st, L, q, out, ln.stage, ln.aux, r->acc);
      else if (r->probe_any) hipLaunchKernelGGL(k_shade_pt<true>, dim3(shade_grid), dim3(kWave), 0, st, L, q, out, ln.stage);
      else hipLaunchKernelGGL(k_shade_pt<false>, dim3(shade_grid), dim3(kWave), 0, st, L, q, out, ln.stage);
    }
    if (S.integrator != IT_VPT && S.integrator != IT_LT) {
      {
        ScopedTimer t(r, st, CASTER_K_SHADOW);
        hipLaunchKernelGGL(k_trace<true>, dim3(walk_grid / kTraceBlockWaves), dim3(kWave * kTraceBlockWaves), 0, st, L, ln.shadow, aux_for(1));
      }
      if (S.integrator == IT_PT && !r->probe_any) {   // (else the probe rays were among the shadow rays, their terms settled)
        {
          ScopedTimer t(r, st, CASTER_K_PROBE);
          hipLaunchKernelGGL(k_trace<false>, dim3(walk_grid / kTraceBlockWaves), dim3(kWave * kTraceBlockWaves), 0, st, L, ln.probe, aux_for(2));
        }
        hipLaunchKernelGGL(k_probe_resolve, dim3(shade_grid), dim3(kWave), 0, st, L, ln.probe);
      }
    }
    if (r->lt_queues) lt_connections();
    // the camera queue also holds slots outside the image / row range: count only live rays
    // (light tracing on queues: its shadow queue and list are empty again, and were no shadow or probe rays)
    hipLaunchKernelGGL(k_roll, dim3(1), dim3(1), 0, st, ln.stats, (const uint32_t*)q.count,
                       r->lt_queues ? (const uint32_t*)nullptr : (const uint32_t*)ln.shadow.count,
                       r->lt_queues ? (const uint32_t*)nullptr : (const uint32_t*)ln.probe.count, q.count, ln.shadow.count,
                       ln.probe.count, b == 0 ? live_rays : 0u, ln.counters + 64, 3u * kWorkShards, r->work_init,
                       r->probe_any ? 1u : 0u);
    cur ^= 1;
  };
  if (!tail) {
    if (S.integrator == IT_AO) bounce(0);
    else for (int b = 0; b < S.maxDepth; ++b) bounce(b);
    ln.batch_first = first; ln.batch_n = n; ln.batch_cur = cur; ln.batch_b = S.maxDepth;
    HIPCHECK(hipGetLastError());
    return CASTER_OK;
  }
  if (S.integrator == IT_VPT || S.integrator == IT_LT) {
    // Boundary crossings do not consume a bounce (pathtracer.cu:1106-1113, :1318-1325), so
    // the number of rounds is data dependent: after maxDepth rounds keep going
    // while the queue is non-empty.
    int b = ln.batch_b;
    // (the queue length is only asked for every few rounds: a round over an empty
    // queue costs a handful of empty launches, a host round trip stalls the stream)
    for (;;) {
      uint32_t left = 0;
      HIPCHECK(hipMemcpyAsync(&left, ln.ext[cur].count, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      HIPCHECK(hipStreamSynchronize(st));
      if (left == 0 || b > S.maxDepth + kMaxCrossings) break;
      for (int k = 0; k < 4; ++k, ++b) bounce(b);
    }
  }
  if (S.integrator == IT_LT) {
    // light paths have splatted their contributions into the film as they went (float atomics,
    // from every stream at once): there is no per-pixel colour to add, and a read-modify-write
    // pass over the film would race with the other streams' splats
    HIPCHECK(hipGetLastError());
    return CASTER_OK;
  }
  HIPCHECK(hipGetLastError());
  return finish ? finish_iterations(r, lane, n, 0, n) : CASTER_OK;
}

// Adds iterations k0 .. k1-1 of the batch of `n` that lane `lane` has traced to the film.
static int finish_iterations(caster_renderer* r, uint32_t lane, uint32_t n, uint32_t k0, uint32_t k1) {
  Lane& ln = r->lanes[lane];
  hipStream_t st = ln.stream;
  // a pixel's samples are added in iteration order whatever stream traced them
  if (r->last_finish && r->last_finish != ln.finish_done) HIPCHECK(hipStreamWaitEvent(st, r->last_finish, 0));
  {
    ScopedTimer t(r, st, CASTER_K_FINISH);
    hipLaunchKernelGGL(k_finish, dim3(r->n_cu * 8), dim3(256), 0, st, ctx_of(r, ln), r->acc, n, k0, k1);
  }
  HIPCHECK(hipEventRecord(ln.finish_done, st));
  r->last_finish = ln.finish_done;
  HIPCHECK(hipGetLastError());
  return CASTER_OK;
}

}  // namespace caster

__global__ void k_film_add(float* acc, const float* add, uint32_t n) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc[i] += add[i];
}

namespace caster {
static int multi_render_async(caster_renderer* g, uint32_t first_iter, uint32_t n_iter, int reset);
static int multi_read_film_device(caster_renderer* g, float* dev, int mode);
template <class Fn> static int for_each_part(caster_renderer* g, Fn fn);
}

extern "C" {

const char* caster_last_error(void) { return g_last_error.c_str(); }

int caster_scene_load(const char* json_path, caster_scene** out) {
  if (!json_path || !out) return fail(CASTER_ERR_ARG, "caster_scene_load: null argument");
  caster_scene* s = new caster_scene();
  std::string err;
  bool ok = false;
  try {
    ok = load_scene(json_path, s->scene, err);
  } catch (const std::exception& e) {   // (a malformed file that asks for more memory than there is, and the like)
    err = std::string("loading the scene failed: ") + e.what();
  } catch (...) {
    err = "loading the scene failed";
  }
  if (!ok) {
    delete s;
    return fail(CASTER_ERR_SCENE, err);
  }
  *out = s;
  return CASTER_OK;
}

static bool gpu_builder(void* ctx, const std::vector<BBox>& boxes, const BBox& root_box, std::vector<Node>& nodes,
                        std::vector<int>& order, std::string& err) {
  return build_bvh_gpu(*(int*)ctx, boxes, root_box, nodes, order, err);
}

int caster_scene_load_device(const char* json_path, int device_id, caster_scene** out) {
  if (!json_path || !out) return fail(CASTER_ERR_ARG, "caster_scene_load_device: null argument");
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || device_id < 0 || device_id >= n_dev)
    return fail(CASTER_ERR_DEVICE, "caster_scene_load_device: no such HIP device");
  caster_scene* s = new caster_scene();
  std::string err;
  bool ok = false;
  try {
    ok = load_scene(json_path, s->scene, err, gpu_builder, &device_id);
  } catch (const std::exception& e) {
    err = std::string("loading the scene failed: ") + e.what();
  } catch (...) {
    err = "loading the scene failed";
  }
  if (!ok) {
    delete s;
    return fail(CASTER_ERR_SCENE, err);
  }
  *out = s;
  return CASTER_OK;
}

int caster_scene_get_bvh(const caster_scene* s, float* boxes, int32_t* meta, int32_t* prim_push_ids) {
  if (!s) return fail(CASTER_ERR_ARG, "caster_scene_get_bvh: null scene");
  const Scene& sc = s->scene;
  for (size_t i = 0; i < sc.nodes.size(); ++i) {
    const Node& n = sc.nodes[i];
    if (boxes) {
      float* b = boxes + 6 * i;
      b[0] = n.lo.x; b[1] = n.lo.y; b[2] = n.lo.z; b[3] = n.hi.x; b[4] = n.hi.y; b[5] = n.hi.z;
    }
    if (meta) { int32_t* m = meta + 4 * i; m[0] = n.second; m[1] = n.leaf; m[2] = n.start; m[3] = n.end; }
  }
  if (prim_push_ids)
    for (size_t i = 0; i < sc.prims.size(); ++i) prim_push_ids[i] = sc.prims[i].push_id;
  return CASTER_OK;
}

void caster_scene_free(caster_scene* s) { delete s; }

int caster_scene_get_info(const caster_scene* s, caster_scene_info* o) {
  if (!s || !o) return fail(CASTER_ERR_ARG, "caster_scene_get_info: null argument");
  const Scene& sc = s->scene;
  o->width = sc.width; o->height = sc.height; o->integrator = sc.integrator; o->max_depth = sc.maxDepth;
  o->n_prims = (int)sc.prims.size(); o->n_nodes = (int)sc.nodes.size(); o->n_lights = (int)sc.lights.size();
  o->n_materials = (int)sc.materials.size(); o->n_media = (int)sc.media.size();
  o->n_textures = (int)sc.textures.size(); o->has_environment = sc.infinite.valid; o->filmic = sc.camera.filmic;
  o->max_dist = sc.maxDist; o->epsilon = sc.epsilon;
  return CASTER_OK;
}

void caster_end(caster_renderer* r) {
  if (!r) return;
  if (!r->parts.empty()) {   // (the group's own buffers are among its first part's allocations)
    for (caster_renderer* p : r->parts) caster_end(p);
    delete r;
    return;
  }
  (void)hipSetDevice(r->device);
  for (Lane& ln : r->lanes) {
    if (ln.stream) { (void)hipStreamSynchronize(ln.stream); (void)hipStreamDestroy(ln.stream); }
    if (ln.finish_done) (void)hipEventDestroy(ln.finish_done);
  }
  for (auto& t : r->pending_events) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
  for (hipEvent_t e : r->free_events) (void)hipEventDestroy(e);
  for (void* p : r->allocs) (void)hipFree(p);
  delete r;
}

int caster_begin_streams(const caster_scene* s, int device_id, int iters_in_flight, int n_streams,
                         caster_renderer** out) {
  if (!s || !out) return fail(CASTER_ERR_ARG, "caster_begin: null argument");
  if (s->scene.prims.empty()) return fail(CASTER_ERR_SCENE, "caster_begin: scene has no primitives");
  for (const Prim& p : s->scene.prims)
    if (p.mat < 0 && s->scene.integrator == IT_PT)
      return fail(CASTER_ERR_SCENE, "caster_begin: a shape without a material needs the vpt integrator");
  caster_renderer* r = new caster_renderer();
  g_out_of_memory = false;
  int rc = begin_impl(s->scene, device_id, iters_in_flight, n_streams, r);
  // Default pools are sized from the memory that is free at that moment; when another renderer took its share
  // in the meantime (several processes on one GPU), the allocation fails and the pools are halved until they fit.
  // (Only a failed allocation is tried again: any other device error is reported as it is.)
  while (rc == CASTER_ERR_DEVICE && g_out_of_memory && iters_in_flight <= 0 && r->iters_in_flight > 1) {
    const int smaller = r->iters_in_flight / 2;
    const int streams = (int)r->lanes.size();
    std::string keep = g_last_error;
    caster_end(r);
    (void)hipGetLastError();   // (the failed allocation is not to be found by the next launch check)
    r = new caster_renderer();
    g_out_of_memory = false;
    rc = begin_impl(s->scene, device_id, smaller, streams, r);
    if (rc == CASTER_ERR_DEVICE && smaller == 1) { g_last_error = keep; break; }
    if (rc == CASTER_ERR_DEVICE) r->iters_in_flight = smaller;   // (so that the next round halves again)
  }
  if (rc) { caster_end(r); return rc; }
  *out = r;
  return CASTER_OK;
}

int caster_begin(const caster_scene* s, int device_id, int iters_in_flight, caster_renderer** out) {
  return caster_begin_streams(s, device_id, iters_in_flight, 0, out);
}

int caster_set_row_stripes(caster_renderer* r, int y0, int y1, int rows, int stride) {
  if (r && !r->parts.empty()) return fail(CASTER_ERR_ARG, "caster_set_row_stripes: a multi-device renderer deals the scanlines itself");
  if (!r || y0 < 0 || y1 > r->S.height || y0 > y1 || rows < 1 || stride < rows)
    return fail(CASTER_ERR_ARG, "caster_set_row_stripes: bad row set");
  HIPCHECK(hipSetDevice(r->device));
  r->ahead_lane = -1;   // (a window traced ahead lives in the pools this call rewrites)
  int rc = sync_lanes(r);   // batches under way were laid out for the previous row set
  if (rc) return rc;
  DevScene& S = r->S;
  S.row_y0 = y0; S.row_y1 = y1; S.row_rows = rows; S.row_stride = stride;
  auto live = [&](int y) { return y >= y0 && y < y1 && (y - y0) % stride < rows; };
  uint32_t n_rows = 0;
  for (int y = 0; y < S.height; ++y) n_rows += live(y) ? 1u : 0u;
  r->live_pixels = n_rows * (uint32_t)S.width;
  const int all_tile_rows = (S.height + 7) / 8;
  if (y0 % 8 == 0 && rows % 8 == 0 && stride % 8 == 0) {
    // slabs of whole tile rows: only those are given path slots
    S.tile_row0 = y0 / 8; S.tile_rows = rows / 8; S.tile_stride = stride / 8;
    int n = 0;
    for (int ty = 0; ty < all_tile_rows; ++ty) n += live(ty * 8) ? 1 : 0;
    S.tiles_y = std::max(n, 1);
  } else if (rows == stride && y0 % 8 == 0) {
    // one band that starts on a tile row: the tile rows it touches
    S.tile_row0 = y0 / 8; S.tile_rows = 1; S.tile_stride = 1;
    S.tiles_y = std::max((y1 - y0 + 7) / 8, 1);
  } else {
    S.tile_row0 = 0; S.tile_rows = 1; S.tile_stride = 1;
    S.tiles_y = all_tile_rows;
  }
  S.slots_per_iter = S.tiles_x * S.tiles_y * 64;
  // the path pool was sized for iters_in_flight iterations of the whole image: a part of the image gets
  // proportionally more iterations per batch, so that its batches keep their size in paths
  long long fit = (long long)r->pool_slots / S.slots_per_iter;
  if (r->S.integrator == IT_BDPT) fit = r->iters_in_flight;   // (its queues only serve the first-hit map)
  r->batch_iters = (int)std::max<long long>(1, std::min<long long>(fit, 1024));
  return CASTER_OK;
}

int caster_set_rows(caster_renderer* r, int y0, int y1) {
  if (!r || y0 < 0 || y1 > r->S.height || y0 > y1) return fail(CASTER_ERR_ARG, "caster_set_rows: bad range");
  return caster_set_row_stripes(r, y0, y1, 1, 1);
}

int caster_render_async(caster_renderer* r, uint32_t first_iter, uint32_t n_iter, int reset) {
  if (!r) return fail(CASTER_ERR_ARG, "caster_render: null renderer");
  if (!r->parts.empty()) return multi_render_async(r, first_iter, n_iter, reset);
  HIPCHECK(hipSetDevice(r->device));
  if (reset) {
    int rc = sync_lanes(r);
    if (rc) return rc;
    HIPCHECK(hipMemsetAsync(r->acc, 0, (size_t)r->S.width * r->S.height * 3 * sizeof(float), r->lanes[0].stream));
    HIPCHECK(hipStreamSynchronize(r->lanes[0].stream));
    r->iterations = 0;
  }
  if (r->live_pixels == 0) {   // a renderer that owns no scanline (a device group larger than the image has slabs)
    r->ahead_lane = -1;
    r->iterations += n_iter;
    return CASTER_OK;
  }
  const uint32_t n_lanes = (uint32_t)r->lanes.size();
  // One iteration per call with a look-ahead window: the window's iterations are traced together the first
  // time one of them is asked for, and each call adds its own iteration to the film.
  const bool splats = r->S.integrator == IT_LT || r->S.integrator == IT_BDPT;
  const bool look = r->lookahead > 1 && n_iter == 1 && !splats;
  if (r->ahead_lane >= 0 && (!look || reset || first_iter != r->ahead_first + r->ahead_done)) r->ahead_lane = -1;
  if (look) {
    if (r->ahead_lane < 0) {
      const uint32_t n = (uint32_t)std::min(r->lookahead, r->batch_iters);
      const uint32_t lane = r->next_lane;
      int rc = enqueue_batch(r, lane, false, first_iter, n);
      if (!rc) rc = enqueue_batch(r, lane, true, 0, 0, false);
      if (rc) return rc;
      r->next_lane = (r->next_lane + 1) % n_lanes;
      r->ahead_lane = (int)lane; r->ahead_first = first_iter; r->ahead_n = n; r->ahead_done = 0;
    }
    int rc = finish_iterations(r, (uint32_t)r->ahead_lane, r->ahead_n, r->ahead_done, r->ahead_done + 1);
    if (rc) return rc;
    if (++r->ahead_done == r->ahead_n) r->ahead_lane = -1;
    r->iterations += 1;
    r->samples += (uint64_t)r->live_pixels;
    return CASTER_OK;
  }
  uint32_t done = 0;
  while (done < n_iter) {
    // one batch per stream: first what needs no host round trip, then the rest in batch order
    uint32_t heads = 0;
    for (; heads < n_lanes && done < n_iter; ++heads) {
      uint32_t n = std::min<uint32_t>(n_iter - done, (uint32_t)r->batch_iters);
      int rc = enqueue_batch(r, (r->next_lane + heads) % n_lanes, false, first_iter + done, n);
      if (rc) return rc;
      done += n;
    }
    for (uint32_t k = 0; k < heads; ++k) {
      int rc = enqueue_batch(r, (r->next_lane + k) % n_lanes, true, 0, 0);
      if (rc) return rc;
    }
    r->next_lane = (r->next_lane + heads) % n_lanes;
  }
  r->iterations += n_iter;
  r->samples += (uint64_t)r->live_pixels * n_iter;
  return CASTER_OK;
}

int caster_set_lookahead(caster_renderer* r, int n_iter) {
  if (!r || n_iter < 0) return fail(CASTER_ERR_ARG, "caster_set_lookahead: bad argument");
  if (!r->parts.empty()) return for_each_part(r, [&](int i) { return caster_set_lookahead(r->parts[(size_t)i], n_iter); });
  r->lookahead = n_iter;
  r->ahead_lane = -1;
  return CASTER_OK;
}

int caster_sync(caster_renderer* r) {
  if (!r) return fail(CASTER_ERR_ARG, "caster_sync: null renderer");
  if (!r->parts.empty()) return for_each_part(r, [&](int i) { return caster_sync(r->parts[(size_t)i]); });
  HIPCHECK(hipSetDevice(r->device));
  int rc = sync_lanes(r);
  if (rc) return rc;
  collect_timers(r);
  return CASTER_OK;
}

int caster_set_profiling(caster_renderer* r, int enable) {
  if (!r) return fail(CASTER_ERR_ARG, "caster_set_profiling: null renderer");
  if (!r->parts.empty()) return for_each_part(r, [&](int i) { return caster_set_profiling(r->parts[(size_t)i], enable); });
  int rc = caster_sync(r);
  if (rc) return rc;
  r->profiling = enable != 0;
  for (int i = 0; i < CASTER_N_KERNEL_CLASSES; ++i) { r->class_ms[i] = 0; r->class_launches[i] = 0; }
  return CASTER_OK;
}

int caster_get_kernel_times(caster_renderer* r, caster_kernel_times* out) {
  if (!r || !out) return fail(CASTER_ERR_ARG, "caster_get_kernel_times: null argument");
  if (!r->parts.empty()) {   // summed over the devices
    for (int k = 0; k < CASTER_N_KERNEL_CLASSES; ++k) { out->ms[k] = 0; out->launches[k] = 0; }
    for (caster_renderer* p : r->parts) {
      caster_kernel_times t;
      int rc = caster_get_kernel_times(p, &t);
      if (rc) return rc;
      for (int k = 0; k < CASTER_N_KERNEL_CLASSES; ++k) { out->ms[k] += t.ms[k]; out->launches[k] += t.launches[k]; }
    }
    return CASTER_OK;
  }
  int rc = caster_sync(r);
  if (rc) return rc;
  for (int i = 0; i < CASTER_N_KERNEL_CLASSES; ++i) { out->ms[i] = r->class_ms[i]; out->launches[i] = r->class_launches[i]; }
  return CASTER_OK;
}

int caster_render(caster_renderer* r, uint32_t first_iter, uint32_t n_iter, int reset) {
  int rc = caster_render_async(r, first_iter, n_iter, reset);
  if (rc) return rc;
  rc = caster_sync(r);
  if (rc) return rc;
  caster_stats st;
  rc = caster_get_stats(r, &st);
  if (rc) return rc;
  if (st.stack_overflow) return fail(CASTER_ERR_OVERFLOW, "BVH traversal stack overflow: image is invalid");
  return CASTER_OK;
}

int caster_read_film_device(caster_renderer* r, float* dev, int mode) {
  if (!r || !dev || mode < 0 || mode > 2) return fail(CASTER_ERR_ARG, "caster_read_film: bad argument");
  if (!r->parts.empty()) return multi_read_film_device(r, dev, mode);
  HIPCHECK(hipSetDevice(r->device));
  int rc = sync_lanes(r);
  if (rc) return rc;
  hipStream_t st = r->lanes[0].stream;
  uint32_t npix = (uint32_t)r->S.width * (uint32_t)r->S.height;
  float div = mode == 2 ? 1.f : (float)(r->iterations ? r->iterations : 1u);
  hipLaunchKernelGGL(k_resolve_film, dim3(r->n_cu * 4), dim3(256), 0, st, (const float*)r->acc, dev, npix, div,
                     mode == 1 ? 1 : 0, r->S.cam.filmic);
  HIPCHECK(hipGetLastError());
  HIPCHECK(hipStreamSynchronize(st));
  return CASTER_OK;
}

int caster_read_film(caster_renderer* r, float* rgb, int mode) {
  if (!r || !rgb) return fail(CASTER_ERR_ARG, "caster_read_film: null argument");
  float* tmp = r->parts.empty() ? r->film_tmp : r->group_sum;   // (a group resolves its summed film in place)
  int rc = caster_read_film_device(r, tmp, mode);
  if (rc) return rc;
  HIPCHECK(hipMemcpy(rgb, tmp, (size_t)r->S.width * r->S.height * 3 * sizeof(float), hipMemcpyDeviceToHost));
  return CASTER_OK;
}

int caster_write_image(const char* path, int width, int height, const float* rgb) {
  if (!path || !rgb || width <= 0 || height <= 0) return fail(CASTER_ERR_ARG, "caster_write_image: bad argument");
  std::string err;
  if (!write_image(path, width, height, rgb, err)) return fail(CASTER_ERR_ARG, err);
  return CASTER_OK;
}

int caster_save_image(caster_renderer* r, const char* path, int mode) {
  if (!r || !path || mode < -1 || mode > 2) return fail(CASTER_ERR_ARG, "caster_save_image: bad argument");
  if (mode < 0) {
    size_t n = strlen(path);
    bool png = n >= 4 && (path[n - 3] == 'p' || path[n - 3] == 'P') && (path[n - 2] == 'n' || path[n - 2] == 'N');
    mode = png ? 1 : 0;
  }
  std::vector<float> host((size_t)r->S.width * r->S.height * 3);
  int rc = caster_read_film(r, host.data(), mode);
  if (rc) return rc;
  return caster_write_image(path, r->S.width, r->S.height, host.data());
}


int caster_film_add_device(caster_renderer* r, const float* sum_device, uint32_t n_iters) {
  if (!r || !sum_device) return fail(CASTER_ERR_ARG, "caster_film_add_device: null argument");
  if (!r->parts.empty()) {   // merged into the first device's film; the iterations count for the whole group
    int rc = caster_film_add_device(r->parts[0], sum_device, 0);
    if (!rc) r->iterations += n_iters;
    return rc;
  }
  HIPCHECK(hipSetDevice(r->device));
  int rc = sync_lanes(r);
  if (rc) return rc;
  hipStream_t st = r->lanes[0].stream;
  uint32_t n = (uint32_t)r->S.width * (uint32_t)r->S.height * 3u;
  hipLaunchKernelGGL(k_film_add, dim3(r->n_cu * 4), dim3(256), 0, st, r->acc, sum_device, n);
  HIPCHECK(hipGetLastError());
  HIPCHECK(hipStreamSynchronize(st));
  r->iterations += n_iters;
  return CASTER_OK;
}

// ---- film checkpoint ------------------------------------------------------------------------------------------
namespace caster {
struct CheckpointHeader {
  char magic[8];       // "CSTRFILM"
  uint32_t version;    // 1
  uint32_t width, height;
  uint32_t iterations; // iterations the sum holds
  int32_t integrator;
  uint32_t reserved[3];
};
static const char kCheckpointMagic[8] = {'C', 'S', 'T', 'R', 'F', 'I', 'L', 'M'};
}  // namespace caster

int caster_checkpoint_save(caster_renderer* r, const char* path) {
  if (!r || !path) return fail(CASTER_ERR_ARG, "caster_checkpoint_save: null argument");
  const size_t n = (size_t)r->S.width * r->S.height * 3;
  std::vector<float> host(n);
  int rc = caster_read_film(r, host.data(), 2);
  if (rc) return rc;
  CheckpointHeader h;
  memset(&h, 0, sizeof(h));
  memcpy(h.magic, kCheckpointMagic, 8);
  h.version = 1; h.width = (uint32_t)r->S.width; h.height = (uint32_t)r->S.height;
  h.iterations = r->iterations; h.integrator = r->S.integrator;
  // written beside the target and renamed over it: a reader never sees half a checkpoint
  const std::string tmp = std::string(path) + ".part";
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return fail(CASTER_ERR_ARG, "caster_checkpoint_save: cannot write " + tmp);
  bool ok = fwrite(&h, sizeof(h), 1, f) == 1 && fwrite(host.data(), sizeof(float), n, f) == n;
  ok = (fclose(f) == 0) && ok;
  if (!ok || rename(tmp.c_str(), path) != 0) {
    remove(tmp.c_str());
    return fail(CASTER_ERR_ARG, std::string("caster_checkpoint_save: writing ") + path + " failed");
  }
  return CASTER_OK;
}

int caster_checkpoint_load(caster_renderer* r, const char* path) {
  if (!r || !path) return fail(CASTER_ERR_ARG, "caster_checkpoint_load: null argument");
  FILE* f = fopen(path, "rb");
  if (!f) return fail(CASTER_ERR_ARG, std::string("caster_checkpoint_load: cannot read ") + path);
  CheckpointHeader h;
  const size_t n = (size_t)r->S.width * r->S.height * 3;
  std::vector<float> host(n);
  bool ok = fread(&h, sizeof(h), 1, f) == 1;
  const bool ours = ok && memcmp(h.magic, kCheckpointMagic, 8) == 0 && h.version == 1;
  const bool fits = ours && h.width == (uint32_t)r->S.width && h.height == (uint32_t)r->S.height && h.integrator == r->S.integrator;
  if (fits) ok = fread(host.data(), sizeof(float), n, f) == n && fgetc(f) == EOF;
  fclose(f);
  if (!ours) return fail(CASTER_ERR_ARG, std::string("caster_checkpoint_load: ") + path + " is not a film checkpoint");
  if (!fits) return fail(CASTER_ERR_ARG, std::string("caster_checkpoint_load: ") + path + " was written for another image size or integrator");
  if (!ok) return fail(CASTER_ERR_ARG, std::string("caster_checkpoint_load: ") + path + " is truncated or holds trailing bytes");
  // an empty film (whatever was traced ahead is dropped with it), then the stored sum and its iteration count
  int rc = caster_render(r, 1, 0, 1);
  if (rc) return rc;
  if (!r->parts.empty() && r->shard_mode == CASTER_SHARD_TILE) {
    // Every part of a group that deals scanlines gets the stored sum of ITS scanlines: a part's film then goes on
    // exactly as it would have without the interruption, and the gathered film is the uninterrupted one bit for bit.
    const size_t row = (size_t)r->S.width * 3;
    for (caster_renderer* p : r->parts) {
      std::vector<float> mine(n, 0.f);
      const DevScene& S = p->S;
      for (int y = 0; y < S.height; ++y)
        if (y >= S.row_y0 && y < S.row_y1 && (y - S.row_y0) % S.row_stride < S.row_rows)
          memcpy(&mine[(size_t)y * row], &host[(size_t)y * row], row * sizeof(float));
      HIPCHECK(hipSetDevice(p->device));
      HIPCHECK(hipMemcpy(p->film_tmp, mine.data(), n * sizeof(float), hipMemcpyHostToDevice));
      rc = caster_film_add_device(p, p->film_tmp, h.iterations);
      if (rc) return rc;
    }
    r->iterations += h.iterations;
    return CASTER_OK;
  }
  caster_renderer* p0 = r->parts.empty() ? r : r->parts[0];
  float* stage = r->parts.empty() ? r->film_tmp : r->group_peer;
  HIPCHECK(hipSetDevice(p0->device));
  HIPCHECK(hipMemcpy(stage, host.data(), n * sizeof(float), hipMemcpyHostToDevice));
  return caster_film_add_device(r, stage, h.iterations);
}

int caster_primary_ids(caster_renderer* r, uint32_t iter, int32_t* ids) {
  if (!r || !ids) return fail(CASTER_ERR_ARG, "caster_primary_ids: null argument");
  if (!r->parts.empty()) return caster_primary_ids(r->parts[0], iter, ids);
  HIPCHECK(hipSetDevice(r->device));
  r->ahead_lane = -1;   // (a window traced ahead lives in the pools this call rewrites)
  int rc = sync_lanes(r);
  if (rc) return rc;
  Lane& ln = r->lanes[0];
  hipStream_t st = ln.stream;
  LaunchCtx L{r->S_full, ln.P, ln.stats};   // the map is of the whole image whatever rows the renderer traces
  const uint32_t total = (uint32_t)r->S_full.slots_per_iter;
  if (r->S.integrator == IT_LT)  // the first surface each light path reaches
    hipLaunchKernelGGL(k_generate_lt<false>, dim3(r->shade_grid), dim3(kWave), 0, st, L, ln.ext[0], iter, 1u,
                       (float*)nullptr, ln.aux, LtConn{});
  else
    hipLaunchKernelGGL(k_generate, dim3((total + kWave - 1) / kWave), dim3(kWave), 0, st, L, ln.ext[0], iter, 1u);
  hipLaunchKernelGGL(k_set_count, dim3(1), dim3(1), 0, st, ln.ext[0].count, total);
  auto reset_work = [&]() {
    hipLaunchKernelGGL(k_roll, dim3(1), dim3(1), 0, st, ln.stats, (const uint32_t*)nullptr,
                       (const uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                       (uint32_t*)nullptr, 0u, ln.counters + 64, 3u * kWorkShards, r->work_init, 0u);
  };
  reset_work();
  hipLaunchKernelGGL(k_trace<false>, dim3(r->walk_grid / kTraceBlockWaves), dim3(kWave * kTraceBlockWaves), 0, st, L, ln.ext[0], ln.aux);
  reset_work();
  hipLaunchKernelGGL(k_gather_ids, dim3(r->n_cu * 4), dim3(256), 0, st, L, ln.ext[0], r->ids);
  hipLaunchKernelGGL(k_set_count, dim3(1), dim3(1), 0, st, ln.ext[0].count, 0u);
  HIPCHECK(hipGetLastError());
  HIPCHECK(hipStreamSynchronize(st));
  HIPCHECK(hipMemcpy(ids, r->ids, (size_t)r->S.width * r->S.height * sizeof(int32_t), hipMemcpyDeviceToHost));
  return CASTER_OK;
}

int caster_trace_rays(caster_renderer* r, uint32_t n, const float* origins, const float* dirs, const float* tmax,
                      int32_t* prim, float* t) {
  if (!r || !origins || !dirs || !tmax || !prim || !t) return fail(CASTER_ERR_ARG, "caster_trace_rays: null argument");
  if (!r->parts.empty()) return caster_trace_rays(r->parts[0], n, origins, dirs, tmax, prim, t);
  const uint32_t cap = r->S.integrator == IT_BDPT ? (uint32_t)r->S.slots_per_iter : r->pool_slots;
  if (n > cap) return fail(CASTER_ERR_ARG, "caster_trace_rays: more rays than the renderer has path slots");
  if (n == 0) return CASTER_OK;
  HIPCHECK(hipSetDevice(r->device));
  r->ahead_lane = -1;   // (a window traced ahead lives in the pools this call rewrites)
  int rc = sync_lanes(r);
  if (rc) return rc;
  Lane& ln = r->lanes[0];
  hipStream_t st = ln.stream;
  LaunchCtx L = ctx_of(r, ln);
  std::vector<float4> o(n), d(n), hit(n);
  std::vector<int> slot(n);
  for (uint32_t i = 0; i < n; ++i) {
    int none = -1;
    float seed;
    memcpy(&seed, &none, 4);   // no hit known beforehand
    o[i] = make_float4(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2], seed);
    d[i] = make_float4(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], tmax[i]);
    slot[i] = (int)i;
  }
  RayQueue& q = ln.ext[0];
  HIPCHECK(hipMemcpy(q.o, o.data(), n * sizeof(float4), hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(q.d, d.data(), n * sizeof(float4), hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(q.slot, slot.data(), n * sizeof(int), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_set_count, dim3(1), dim3(1), 0, st, q.count, n);
  auto reset_work = [&]() {
    hipLaunchKernelGGL(k_roll, dim3(1), dim3(1), 0, st, ln.stats, (const uint32_t*)nullptr,
                       (const uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                       (uint32_t*)nullptr, 0u, ln.counters + 64, 3u * kWorkShards, r->work_init, 0u);
  };
  reset_work();
  {
    ScopedTimer timer(r, st, CASTER_K_EXTEND);   // (with profiling on, the launch shows in caster_get_kernel_times)
    hipLaunchKernelGGL(k_trace<false>, dim3(r->walk_grid / kTraceBlockWaves), dim3(kWave * kTraceBlockWaves), 0, st, L, q, ln.aux);
  }
  reset_work();
  hipLaunchKernelGGL(k_set_count, dim3(1), dim3(1), 0, st, q.count, 0u);
  HIPCHECK(hipGetLastError());
  HIPCHECK(hipStreamSynchronize(st));
  HIPCHECK(hipMemcpy(hit.data(), q.hit, n * sizeof(float4), hipMemcpyDeviceToHost));
  for (uint32_t i = 0; i < n; ++i) {
    memcpy(&prim[i], &hit[i].y, 4);
    t[i] = prim[i] < 0 ? tmax[i] : hit[i].x;   // (a ray that found nothing reports the bound it came with)
  }
  return CASTER_OK;
}

// caster_trace_rays_static: device-side conversion between the caller's arrays and the queue layout
__global__ void k_pack_rays(uint32_t n, const float* origins, const float* dirs, const float* tmax, RayQueue q) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    q.o[i] = make_float4(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2], __int_as_float(-1));  // no hit known beforehand
    q.d[i] = make_float4(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], tmax[i]);
    q.slot[i] = (int)i;
  }
}
__global__ void k_unpack_hits(uint32_t n, const float4* hit, const float* tmax, int32_t* prim, float* t) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float4 h = hit[i];
    prim[i] = __float_as_int(h.y);
    t[i] = __float_as_int(h.y) < 0 ? tmax[i] : h.x;   // (a ray that found nothing reports the bound it came with)
  }
}

int caster_trace_rays_static(caster_renderer* r, uint32_t n, const float* origins_dev, const float* dirs_dev,
                             const float* tmax_dev, int32_t* prim_dev, float* t_dev) {
  if (!r || !origins_dev || !dirs_dev || !tmax_dev || !prim_dev || !t_dev)
    return fail(CASTER_ERR_ARG, "caster_trace_rays_static: null argument");
  if (!r->parts.empty()) return caster_trace_rays_static(r->parts[0], n, origins_dev, dirs_dev, tmax_dev, prim_dev, t_dev);
  if (n > kStaticMaxRays)
    return fail(CASTER_ERR_ARG, "caster_trace_rays_static: more than CASTER_STATIC_MAX_RAYS (" + std::to_string(kStaticMaxRays) +
                                    ") rays in one call");
  if (n == 0) return CASTER_OK;
  if (!r->static_scratch) return fail(CASTER_ERR_DEVICE, "caster_trace_rays_static: no static scratch on this device");
  HIPCHECK(hipSetDevice(r->device));
  int rc = sync_lanes(r);
  if (rc) return rc;
  Lane& ln = r->lanes[0];
  hipStream_t st = ln.stream;
  LaunchCtx L = ctx_of(r, ln);
  // the queue and its counters, carved out of this device's copy of the scratch array
  unsigned char* base = (unsigned char*)r->static_scratch;
  RayQueue q;
  q.o = (float4*)base;
  q.d = q.o + kStaticMaxRays;
  q.hit = q.d + kStaticMaxRays;
  q.slot = (int*)(q.hit + kStaticMaxRays);
  uint32_t* counters = (uint32_t*)(q.slot + kStaticMaxRays);
  q.count = counters;
  WalkAux aux = ln.aux;          // (the stack spill columns are the renderer's, allocated by caster_begin)
  aux.work = counters + 64;
  // every call on a GPU shares the array, which still holds the previous call's contents: the
  // queue entries [0, n) are rewritten below, the length and the work counters are set here
  HIPCHECK(hipMemsetAsync(counters, 0, kStaticCounterWords * sizeof(uint32_t), st));
  const int grid = (int)std::min<uint32_t>((n + 255) / 256, (uint32_t)r->n_cu * 8);
  hipLaunchKernelGGL(k_pack_rays, dim3(grid), dim3(256), 0, st, n, origins_dev, dirs_dev, tmax_dev, q);
  hipLaunchKernelGGL(k_set_count, dim3(1), dim3(1), 0, st, q.count, n);
  hipLaunchKernelGGL(k_trace<false>, dim3(r->walk_grid / kTraceBlockWaves), dim3(kWave * kTraceBlockWaves), 0, st, L, q, aux);
  hipLaunchKernelGGL(k_unpack_hits, dim3(grid), dim3(256), 0, st, n, (const float4*)q.hit, tmax_dev, prim_dev, t_dev);
  HIPCHECK(hipGetLastError());
  HIPCHECK(hipStreamSynchronize(st));
  return CASTER_OK;
}

int caster_eval_function(caster_renderer* r, int fn, uint32_t n, const float* in, float* out) {
  static const int in_w[CASTER_N_FN] = {16, 16, 6, 3, 8}, out_w[CASTER_N_FN] = {7, 4, 11, 6, 15};
  if (!r || !in || !out || fn < 0 || fn >= CASTER_N_FN) return fail(CASTER_ERR_ARG, "caster_eval_function: bad argument");
  if (!r->parts.empty()) return caster_eval_function(r->parts[0], fn, n, in, out);
  if (n == 0) return CASTER_OK;
  HIPCHECK(hipSetDevice(r->device));
  int rc = sync_lanes(r);
  if (rc) return rc;
  float *din = nullptr, *dout = nullptr;
  HIPCHECK(hipMalloc((void**)&din, (size_t)n * in_w[fn] * sizeof(float)));
  if (hipMalloc((void**)&dout, (size_t)n * out_w[fn] * sizeof(float)) != hipSuccess) {
    (void)hipFree(din);
    return fail(CASTER_ERR_DEVICE, "caster_eval_function: out of device memory");
  }
  hipStream_t st = r->lanes[0].stream;
  hipError_t e = hipMemcpy(din, in, (size_t)n * in_w[fn] * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    const dim3 grid((n + 63) / 64 < 4096 ? (n + 63) / 64 : 4096), block(64);
    switch (fn) {
      case CASTER_FN_SAMPLE_BSDF: hipLaunchKernelGGL(k_hook_sample_bsdf, grid, block, 0, st, r->S, n, (const float*)din, dout); break;
      case CASTER_FN_EVAL_BSDF: hipLaunchKernelGGL(k_hook_eval_bsdf, grid, block, 0, st, r->S, n, (const float*)din, dout); break;
      case CASTER_FN_SAMPLE_LIGHT: hipLaunchKernelGGL(k_hook_sample_light, grid, block, 0, st, r->S, n, (const float*)din, dout); break;
      case CASTER_FN_CAMERA_RAY: hipLaunchKernelGGL(k_hook_camera_ray, grid, block, 0, st, r->S, n, (const float*)din, dout); break;
      default: hipLaunchKernelGGL(k_hook_hit_record, grid, block, 0, st, r->S, n, (const float*)din, dout); break;
    }
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e == hipSuccess) e = hipMemcpy(out, dout, (size_t)n * out_w[fn] * sizeof(float), hipMemcpyDeviceToHost);
  (void)hipFree(din);
  (void)hipFree(dout);
  if (e != hipSuccess) return fail(CASTER_ERR_DEVICE, std::string("caster_eval_function: ") + hipGetErrorString(e));
  return CASTER_OK;
}

int caster_get_stats(caster_renderer* r, caster_stats* out) {
  if (!r || !out) return fail(CASTER_ERR_ARG, "caster_get_stats: null argument");
  if (!r->parts.empty()) {   // summed over the devices
    memset(out, 0, sizeof(*out));
    for (caster_renderer* p : r->parts) {
      caster_stats st;
      int rc = caster_get_stats(p, &st);
      if (rc) return rc;
      for (int i = 0; i < 4; ++i) out->rays[i] += st.rays[i];
      out->samples += st.samples; out->invalid += st.invalid; out->stack_overflow += st.stack_overflow;
    }
    return CASTER_OK;
  }
  HIPCHECK(hipSetDevice(r->device));
  int rc = sync_lanes(r);
  if (rc) return rc;
  for (int i = 0; i < 4; ++i) out->rays[i] = 0;
  out->invalid = 0;
  out->stack_overflow = 0;
  for (Lane& ln : r->lanes) {
    Stats st;
    HIPCHECK(hipMemcpy(&st, ln.stats, sizeof(Stats), hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; ++i) out->rays[i] += st.rays[i];
    out->invalid += st.invalid;
    out->stack_overflow += st.stack_overflow;
  }
  out->samples = r->samples;
  return CASTER_OK;
}

int caster_get_walk_counts(caster_renderer* r, uint64_t* out4) {
  if (!r || !out4) return fail(CASTER_ERR_ARG, "caster_get_walk_counts: null argument");
  if (!r->parts.empty()) {
    for (int i = 0; i < 4; ++i) out4[i] = 0;
    for (caster_renderer* p : r->parts) {
      uint64_t c[4];
      int rc = caster_get_walk_counts(p, c);
      if (rc) return rc;
      for (int i = 0; i < 4; ++i) out4[i] += c[i];
    }
    return CASTER_OK;
  }
  HIPCHECK(hipSetDevice(r->device));
  int rc = sync_lanes(r);
  if (rc) return rc;
  for (int i = 0; i < 4; ++i) out4[i] = 0;
  for (Lane& ln : r->lanes) {
    Stats st;
    HIPCHECK(hipMemcpy(&st, ln.stats, sizeof(Stats), hipMemcpyDeviceToHost));
    out4[0] += st.node_fetches[0]; out4[1] += st.prim_tests[0];
    out4[2] += st.node_fetches[1]; out4[3] += st.prim_tests[1];
  }
  return CASTER_OK;
}

int caster_reset_stats(caster_renderer* r) {
  if (!r) return fail(CASTER_ERR_ARG, "caster_reset_stats: null renderer");
  if (!r->parts.empty()) return for_each_part(r, [&](int i) { return caster_reset_stats(r->parts[(size_t)i]); });
  HIPCHECK(hipSetDevice(r->device));
  int rc = sync_lanes(r);
  if (rc) return rc;
  for (Lane& ln : r->lanes) HIPCHECK(hipMemset(ln.stats, 0, sizeof(Stats)));
  r->samples = 0;
  return CASTER_OK;
}

uint32_t caster_iterations(const caster_renderer* r) { return r ? r->iterations : 0; }
int caster_iters_in_flight(const caster_renderer* r) { return r ? r->iters_in_flight : 0; }

#ifdef CASTER_DEBUG_SLOW_RAYS
// diagnosis builds only: the rays k_trace recorded as slow (256 x 12 floats) and how many there were
int caster_debug_slow_rays(float* out, unsigned int* count) {
  HIPCHECK(hipDeviceSynchronize());
  HIPCHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_slow), sizeof(float) * 256 * 12));
  HIPCHECK(hipMemcpyFromSymbol(count, HIP_SYMBOL(g_slow_count), sizeof(unsigned int)));
  return CASTER_OK;
}
#endif

#ifdef CASTER_WALK_STATS
// tuning builds only: copies and clears the k_trace step counters
int caster_debug_walk_stats(unsigned long long* out32) {
  HIPCHECK(hipDeviceSynchronize());
  HIPCHECK(hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_walk_stats), 32 * sizeof(unsigned long long)));
  unsigned long long zero[32] = {0};
  HIPCHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_walk_stats), zero, sizeof(zero)));
  return CASTER_OK;
}
#endif

}  // extern "C"

// ---- several devices behind one handle -------------------------------------------------------------------
namespace caster {

// Runs fn(part index) for every part of the group, each on a host thread of its own (HIP's current device is
// per thread, and a call that waits for its device -- caster_sync, the data-dependent rounds of vpt and lt --
// must not hold up the other devices); returns the first failure.
template <class Fn>
static int for_each_part(caster_renderer* g, Fn fn) {
  const size_t n = g->parts.size();
  std::vector<int> rc(n, CASTER_OK);
  std::vector<std::string> msg(n);
  std::vector<std::thread> th;
  th.reserve(n);
  for (size_t i = 0; i < n; ++i)
    th.emplace_back([&, i]() {
      try {
        rc[i] = fn((int)i);
        if (rc[i]) msg[i] = g_last_error;   // (the message is the worker thread's)
      } catch (const std::exception& e) {   // (nothing may leave a thread, or the C boundary)
        rc[i] = CASTER_ERR_DEVICE;
        msg[i] = e.what();
      }
    });
  for (auto& t : th) t.join();
  for (size_t i = 0; i < n; ++i)
    if (rc[i]) return fail(rc[i], "device " + std::to_string(g->parts[i]->device) + ": " + msg[i]);
  return CASTER_OK;
}

// Iterations [first, first + n) as part `k` of `world` renders them under spp sharding: iteration i (counted
// from 1) lies in block (i - 1) / block, and the blocks are dealt round-robin by that absolute number, as
// sharding.spp_blocks deals them to ranks -- so who renders an iteration does not depend on how the caller
// cuts its calls, and a caller that asks for one iteration per call still rotates over the devices.
static void spp_blocks_of(uint32_t first, uint32_t n, uint32_t block, int k, int world, std::vector<std::pair<uint32_t, uint32_t>>& out) {
  const uint64_t end = (uint64_t)first + n;
  uint64_t it = first;
  while (it < end) {
    const uint64_t b = (it > 0 ? it - 1 : 0) / block;
    const uint64_t block_end = std::min<uint64_t>(end, (b + 1) * (uint64_t)block + 1);
    if ((int)(b % (uint64_t)world) == k) out.push_back({(uint32_t)it, (uint32_t)(block_end - it)});
    it = block_end;
  }
}

}  // namespace caster

extern "C" {

int caster_begin_multi(const caster_scene* s, const int* device_ids, int n_dev, int iters_in_flight, int n_streams, int shard_mode,
                       caster_renderer** out) {
  if (!s || !device_ids || !out || n_dev < 1 || n_dev > 64 || (shard_mode != CASTER_SHARD_TILE && shard_mode != CASTER_SHARD_SPP))
    return fail(CASTER_ERR_ARG, "caster_begin_multi: bad argument");
  caster_renderer* g = new caster_renderer();
  g->shard_mode = shard_mode;
  g->parts.assign((size_t)n_dev, nullptr);
  // (CASTER_MULTI_ITERS="4,2,..." in the environment gives the parts these iterations in flight instead: a test
  // knob that stands in for devices with different amounts of free memory)
  std::vector<int> part_iters((size_t)n_dev, iters_in_flight);
  if (const char* e = getenv("CASTER_MULTI_ITERS")) {
    size_t k = 0;
    for (const char* p = e; *p && k < part_iters.size(); ++k) {
      part_iters[k] = atoi(p);
      while (*p && *p != ',') ++p;
      if (*p == ',') ++p;
    }
  }
  int rc = for_each_part(g, [&](int i) { return caster_begin_streams(s, device_ids[i], part_iters[(size_t)i], n_streams, &g->parts[(size_t)i]); });
  if (!rc && shard_mode == CASTER_SHARD_TILE) {
    // slabs of 8 scanlines dealt round-robin: part k of G traces the scanlines y >= 8k with (y - 8k) % 8G < 8
    const int h = s->scene.height;
    rc = for_each_part(g, [&](int i) { return caster_set_row_stripes(g->parts[(size_t)i], std::min(8 * i, h), h, 8, 8 * n_dev); });
  }
  if (!rc) {
    caster_renderer* p0 = g->parts[0];
    g->device = p0->device;
    g->S = p0->S_full;
    g->iters_in_flight = p0->iters_in_flight;
    // One block size for the whole group: the parts size their default pools from the memory free on their
    // device at that moment, so their batches may differ, and parts that cut the iterations into blocks of
    // different sizes would render some iterations twice and others never.
    g->batch_iters = p0->batch_iters;
    for (caster_renderer* p : g->parts) g->batch_iters = std::min(g->batch_iters, p->batch_iters);
    if (hipSetDevice(p0->device) != hipSuccess) rc = fail(CASTER_ERR_DEVICE, "caster_begin_multi: hipSetDevice failed");
    const size_t n = (size_t)g->S.width * g->S.height * 3;
    if (!rc) rc = dev_alloc(p0, &g->group_sum, n);
    if (!rc) rc = dev_alloc(p0, &g->group_peer, n);
  }
  if (rc) {
    std::string keep = g_last_error;
    caster_end(g);
    g_last_error = keep;
    return rc;
  }
  *out = g;
  return CASTER_OK;
}

}  // extern "C"

namespace caster {

static int multi_render_async(caster_renderer* g, uint32_t first_iter, uint32_t n_iter, int reset) {
  const int world = (int)g->parts.size();
  int rc = for_each_part(g, [&](int i) {
    caster_renderer* p = g->parts[(size_t)i];
    if (g->shard_mode == CASTER_SHARD_TILE) return caster_render_async(p, first_iter, n_iter, reset);
    std::vector<std::pair<uint32_t, uint32_t>> blocks;
    spp_blocks_of(first_iter, n_iter, (uint32_t)std::max(g->batch_iters, 1), i, world, blocks);
    if (blocks.empty()) return reset ? caster_render_async(p, first_iter, 0, 1) : (int)CASTER_OK;
    int r1 = CASTER_OK;
    for (size_t b = 0; b < blocks.size() && !r1; ++b) r1 = caster_render_async(p, blocks[b].first, blocks[b].second, b == 0 ? reset : 0);
    return r1;
  });
  if (rc) return rc;
  if (reset) g->iterations = 0;
  g->iterations += n_iter;
  return CASTER_OK;
}

// The films of all parts, added up on the first part's device (disjoint scanlines under tile sharding: an
// exact gather; disjoint iterations under spp sharding: a sum), then resolved as one film.
static int multi_read_film_device(caster_renderer* g, float* dev, int mode) {
  caster_renderer* p0 = g->parts[0];
  const size_t n = (size_t)g->S.width * g->S.height * 3;
  int rc = for_each_part(g, [&](int i) {
    caster_renderer* p = g->parts[(size_t)i];
    return caster_read_film_device(p, i == 0 ? g->group_sum : p->film_tmp, 2);   // the raw sums
  });
  if (rc) return rc;
  HIPCHECK(hipSetDevice(p0->device));
  hipStream_t st = p0->lanes[0].stream;
  for (size_t i = 1; i < g->parts.size(); ++i) {
    caster_renderer* p = g->parts[i];
    HIPCHECK(hipMemcpyPeer(g->group_peer, p0->device, p->film_tmp, p->device, n * sizeof(float)));
    hipLaunchKernelGGL(k_film_add, dim3(p0->n_cu * 4), dim3(256), 0, st, g->group_sum, (const float*)g->group_peer, (uint32_t)n);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(st));
  }
  float div = mode == 2 ? 1.f : (float)(g->iterations ? g->iterations : 1u);
  hipLaunchKernelGGL(k_resolve_film, dim3(p0->n_cu * 4), dim3(256), 0, st, (const float*)g->group_sum, dev, (uint32_t)(n / 3), div,
                     mode == 1 ? 1 : 0, g->S.cam.filmic);
  HIPCHECK(hipGetLastError());
  HIPCHECK(hipStreamSynchronize(st));
  return CASTER_OK;
}

}  // namespace caster
