// Host-side scene description: what a scene file turns into before it is
// handed to a renderer.  Mirrors the data the reference keeps in `Scene`,
// `GlobalConfig` and `BVH` (CUDA-PathTracer/tracer/scene.h:24-82,
// parsescene.h:8-24, bvh.h:19-48) with plain structs instead of the 176-byte
// tagged union.
#pragma once
#include <stdint.h>
#include <cstdint>
#include <string>
#include <vector>
#include "../fpspec.h"

namespace caster {

struct f2 { float x, y; };
struct f3 { float x, y, z; };

inline f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
inline f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
inline f3 operator*(float s, f3 a) { return f3{a.x * s, a.y * s, a.z * s}; }
inline float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline f3 cross3(f3 a, f3 b) {
  return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline f3 normalize3(f3 v) { return v * fps::rsqrt_(dot3(v, v)); }

// Values match the reference enums (primitive.h:7-11, material.h:6-13,
// medium.h:131-134, scene.h:13-22).
enum GeomType { GT_TRIANGLE = 0, GT_LINES = 1, GT_SPHERE = 2 };
enum MaterialType {
  MT_LAMBERTIAN = 0, MT_MIRROR, MT_DIELECTRIC, MT_ROUGHDIELECTRIC, MT_ROUGHCONDUCTOR, MT_SUBSTRATE
};
enum MediumType { MED_HOMOGENEOUS = 0, MED_HETEROGENEOUS = 1 };
enum IntegratorType { IT_AO = 0, IT_PT, IT_VPT, IT_LT, IT_BDPT, IT_MLT, IT_SPPM, IT_IR };

struct Vertex { f3 v, n; f2 uv; f3 t; };  // Triangle.h:8-13

// One primitive.  Triangle: v[0..2].  Sphere: p0 = centre, w0 = radius.
// Line: p0, p1, w0, w1.
struct Prim {
  int type;
  Vertex v[3];
  f3 p0, p1;
  float w0, w1;
  int mat;      // -1: index-matched medium boundary
  int light;    // index into Scene::lights or -1
  int med_in, med_out;
  int push_id;  // position in scene-file order (scene[] then light[])
};

struct BBox {
  f3 lo{INFINITY, INFINITY, INFINITY}, hi{-INFINITY, -INFINITY, -INFINITY};
  void expand(f3 p) {
    lo = f3{fps::fmin_(lo.x, p.x), fps::fmin_(lo.y, p.y), fps::fmin_(lo.z, p.z)};
    hi = f3{fps::fmax_(hi.x, p.x), fps::fmax_(hi.y, p.y), fps::fmax_(hi.z, p.z)};
  }
  void expand(const BBox& b) {
    lo = f3{fps::fmin_(b.lo.x, lo.x), fps::fmin_(b.lo.y, lo.y), fps::fmin_(b.lo.z, lo.z)};
    hi = f3{fps::fmax_(b.hi.x, hi.x), fps::fmax_(b.hi.y, hi.y), fps::fmax_(b.hi.z, hi.z)};
  }
  f3 centre() const { return (lo + hi) * 0.5f; }
  float area() const {
    f3 d = hi - lo;
    return 2.f * (d.x * d.y + d.y * d.z + d.z * d.x);
  }
};

// Pre-order flattened binary node (bvh.h:19-29): first child is index+1.
struct Node {
  f3 lo, hi;
  int second;  // index of the second child, -1 for a leaf
  int leaf;
  int start, end;  // inclusive primitive range of a leaf, -1/-1 otherwise
};

struct Material {  // material.h:15-23
  int type;
  float alphaU, alphaV;
  float insideIOR, outsideIOR;
  f3 k, eta;
  f3 diffuse, specular;
  int tex;
};

struct Texture { int w = 0, h = 0; std::vector<uint8_t> rgba; };

struct Medium {  // medium.h:7-10, 41-49, 136-139
  int type;
  float g;
  f3 sigmaA, sigmaS, sigmaT;
  // heterogeneous only: density grid over the box [p0, p1], z-major (medium.h:124-128)
  int nx = 0, ny = 0, nz = 0;
  std::vector<float> density;
  float invMaxDensity = 0.f;
  f3 p0{0, 0, 0}, p1{0, 0, 0};
  int iterMax = 1000;
};

struct Light {  // area.h:5-10; the triangle is prims[prim]
  f3 radiance;
  int prim;
  int medium;
};

struct Camera {  // Camera.h:6-43
  f3 position, u, v, w;
  float res_x, res_y;
  float distance, fov, apertureRadius, focalDistance;
  int filmic, environment, medium;
  float half_w, half_h;   // `width`, `height` in the reference
  float p2s_x, p2s_y;     // pixel2screen
  float ratio, area;
};

struct Infinite {  // infinite.h:5-12
  int valid = 0;
  int w = 0, h = 0;
  std::vector<f3> data;
  f3 centre{0, 0, 0};
  float radius = 0.f;
  f3 u{1, 0, 0}, v{0, 1, 0}, w_{0, 0, 1};
};

constexpr float kFarFactor = 16384.f;
struct Scene {
  int width = 512, height = 512;
  float epsilon = 0.001f;
  int integrator = IT_PT;
  int maxDepth = 5;
  float maxDist = 0.5f;
  Camera camera{};
  std::vector<Material> materials;
  std::vector<Texture> textures;
  std::vector<Medium> media;
  bool has_heterogeneous = false;  // some medium is a density grid: transmittance then consumes random numbers
  std::vector<Prim> prims;   // canonical (BVH leaf) order; hit ids index this
  std::vector<Node> nodes;
  BBox root_box;
  // Hits farther along a ray than this are not looked for (kFarFactor times the diagonal of the scene's bounds; infinite
  // for a scene without primitives).  The reference looks as far as +infinity (ray.h:12).  Real geometry lies within
  // one diagonal of any point of the scene; what the triangle test reports at 10^8 diagonals is a rounding artefact
  // of a ray all but parallel to a large triangle, and a path that goes on from there sends rays whose origin leaves
  // single precision no room to tell one box of the tree from another: each such ray visits every node and every
  // primitive (on any implementation, the reference's included) -- hundreds of milliseconds for one ray.
  float far_t = INFINITY;
  std::vector<Light> lights;
  std::vector<float> light_cdf;  // lights.size()+1 (+1 with an environment light)
  Infinite infinite;
};

// Parses `path` (reference schema, parsescene.cpp:46-598), loads meshes and
// textures relative to it, builds the BVH and the light CDF (scene.h:48-81).
// Returns false and fills `err` on failure.
// A BVH builder other than the host one (the GPU build): fills `nodes` (pre-order) and `order`
// (primitive indices leaf by leaf) from the primitive boxes; must deliver what build_bvh_host does.
typedef bool (*BvhBuilder)(void* ctx, const std::vector<BBox>& boxes, const BBox& root_box, std::vector<Node>& nodes,
                           std::vector<int>& order, std::string& err);
bool load_scene(const char* path, Scene& out, std::string& err, BvhBuilder builder = nullptr, void* builder_ctx = nullptr);

// bvh.cpp
BBox prim_bbox(const Prim& p);
void build_bvh(std::vector<Prim>& prims, std::vector<Node>& nodes, BBox& root_box);
// its three steps
void prim_boxes(const std::vector<Prim>& prims, std::vector<BBox>& boxes, BBox& root_box);
void build_bvh_host(const std::vector<BBox>& boxes, const BBox& root_box, std::vector<Node>& nodes, std::vector<int>& order);
void apply_prim_order(std::vector<Prim>& prims, const std::vector<int>& order);

// mesh_io.cpp
struct RawMesh {
  std::vector<f3> pos, nrm;
  std::vector<f2> uv;        // empty when the file has none
  std::vector<int> idx;      // 3 per triangle, into pos/nrm/uv
};
bool load_mesh_file(const std::string& path, RawMesh& out, std::string& err);
bool load_texture_file(const std::string& path, Texture& out, std::string& err);
bool load_envmap_file(const std::string& path, Infinite& out, std::string& err);
// image_out.cpp: writes W*H*3 floats (row 0 = bottom scanline) as .png (8-bit, clamped), .exr (half B,G,R) or .pfm
bool write_image(const std::string& path, int w, int h, const float* rgb, std::string& err);
uint16_t float_to_half(float f);

}  // namespace caster
