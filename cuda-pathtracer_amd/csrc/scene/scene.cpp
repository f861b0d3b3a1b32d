// Scene-file loader: same schema, defaults and push order as the reference
// parser (CUDA-PathTracer/tracer/parsescene.cpp:46-617), mesh processing as
// mesh.cpp:29-112, camera as Camera.h:31-43,128-134, light distribution as
// scene.h:48-81.  Out-of-scope keys (bssrdf materials, the lt/bdpt/mlt/sppm/ir
// integrators) are rejected with an error instead of
// being silently mis-rendered.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include "json.hpp"
#include "scene.h"

namespace caster {

namespace {

using cjson::Value;

// Column-major 4x4, m[c][r], as glm stores it.
struct Mat4 {
  float m[4][4];
  static Mat4 identity() {
    Mat4 r;
    memset(r.m, 0, sizeof(r.m));
    for (int i = 0; i < 4; ++i) r.m[i][i] = 1.f;
    return r;
  }
};

Mat4 mul(const Mat4& a, const Mat4& b) {
  Mat4 r;
  for (int c = 0; c < 4; ++c)
    for (int row = 0; row < 4; ++row) {
      float s = 0.f;
      for (int k = 0; k < 4; ++k) s += a.m[k][row] * b.m[c][k];
      r.m[c][row] = s;
    }
  return r;
}

void xform(const Mat4& a, const float v[4], float out[4]) {
  for (int r = 0; r < 4; ++r)
    out[r] = (a.m[0][r] * v[0] + a.m[1][r] * v[1]) + (a.m[2][r] * v[2] + a.m[3][r] * v[3]);
}

Mat4 translate(f3 t) {
  Mat4 r = Mat4::identity();
  r.m[3][0] = t.x; r.m[3][1] = t.y; r.m[3][2] = t.z;
  return r;
}
Mat4 scale(f3 s) {
  Mat4 r = Mat4::identity();
  r.m[0][0] = s.x; r.m[1][1] = s.y; r.m[2][2] = s.z;
  return r;
}
// Rotation about a principal axis (0=x,1=y,2=z) by `deg` degrees.
Mat4 rotate_axis(float deg, int axis) {
  float a = deg * 0.01745329251994329576923690768489f;
  float c = cosf(a), s = sinf(a);
  Mat4 r = Mat4::identity();
  int i = (axis + 1) % 3, j = (axis + 2) % 3;
  r.m[i][i] = c; r.m[j][j] = c;
  r.m[i][j] = s; r.m[j][i] = -s;
  return r;
}
Mat4 trs_of(f3 t, f3 rot, f3 s) {
  Mat4 r = mul(mul(rotate_axis(rot.x, 0), rotate_axis(rot.y, 1)), rotate_axis(rot.z, 2));
  return mul(mul(translate(t), r), scale(s));
}

// Inverse of a general 4x4 (cofactor expansion, double accumulators).
bool inverse(const Mat4& a, Mat4& out) {
  double m[16], inv[16];
  for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) m[c * 4 + r] = a.m[c][r];
  inv[0] = m[5]*m[10]*m[15] - m[5]*m[11]*m[14] - m[9]*m[6]*m[15] + m[9]*m[7]*m[14] + m[13]*m[6]*m[11] - m[13]*m[7]*m[10];
  inv[4] = -m[4]*m[10]*m[15] + m[4]*m[11]*m[14] + m[8]*m[6]*m[15] - m[8]*m[7]*m[14] - m[12]*m[6]*m[11] + m[12]*m[7]*m[10];
  inv[8] = m[4]*m[9]*m[15] - m[4]*m[11]*m[13] - m[8]*m[5]*m[15] + m[8]*m[7]*m[13] + m[12]*m[5]*m[11] - m[12]*m[7]*m[9];
  inv[12] = -m[4]*m[9]*m[14] + m[4]*m[10]*m[13] + m[8]*m[5]*m[14] - m[8]*m[6]*m[13] - m[12]*m[5]*m[10] + m[12]*m[6]*m[9];
  inv[1] = -m[1]*m[10]*m[15] + m[1]*m[11]*m[14] + m[9]*m[2]*m[15] - m[9]*m[3]*m[14] - m[13]*m[2]*m[11] + m[13]*m[3]*m[10];
  inv[5] = m[0]*m[10]*m[15] - m[0]*m[11]*m[14] - m[8]*m[2]*m[15] + m[8]*m[3]*m[14] + m[12]*m[2]*m[11] - m[12]*m[3]*m[10];
  inv[9] = -m[0]*m[9]*m[15] + m[0]*m[11]*m[13] + m[8]*m[1]*m[15] - m[8]*m[3]*m[13] - m[12]*m[1]*m[11] + m[12]*m[3]*m[9];
  inv[13] = m[0]*m[9]*m[14] - m[0]*m[10]*m[13] - m[8]*m[1]*m[14] + m[8]*m[2]*m[13] + m[12]*m[1]*m[10] - m[12]*m[2]*m[9];
  inv[2] = m[1]*m[6]*m[15] - m[1]*m[7]*m[14] - m[5]*m[2]*m[15] + m[5]*m[3]*m[14] + m[13]*m[2]*m[7] - m[13]*m[3]*m[6];
  inv[6] = -m[0]*m[6]*m[15] + m[0]*m[7]*m[14] + m[4]*m[2]*m[15] - m[4]*m[3]*m[14] - m[12]*m[2]*m[7] + m[12]*m[3]*m[6];
  inv[10] = m[0]*m[5]*m[15] - m[0]*m[7]*m[13] - m[4]*m[1]*m[15] + m[4]*m[3]*m[13] + m[12]*m[1]*m[7] - m[12]*m[3]*m[5];
  inv[14] = -m[0]*m[5]*m[14] + m[0]*m[6]*m[13] + m[4]*m[1]*m[14] - m[4]*m[2]*m[13] - m[12]*m[1]*m[6] + m[12]*m[2]*m[5];
  inv[3] = -m[1]*m[6]*m[11] + m[1]*m[7]*m[10] + m[5]*m[2]*m[11] - m[5]*m[3]*m[10] - m[9]*m[2]*m[7] + m[9]*m[3]*m[6];
  inv[7] = m[0]*m[6]*m[11] - m[0]*m[7]*m[10] - m[4]*m[2]*m[11] + m[4]*m[3]*m[10] + m[8]*m[2]*m[7] - m[8]*m[3]*m[6];
  inv[11] = -m[0]*m[5]*m[11] + m[0]*m[7]*m[9] + m[4]*m[1]*m[11] - m[4]*m[3]*m[9] - m[8]*m[1]*m[7] + m[8]*m[3]*m[5];
  inv[15] = m[0]*m[5]*m[10] - m[0]*m[6]*m[9] - m[4]*m[1]*m[10] + m[4]*m[2]*m[9] + m[8]*m[1]*m[6] - m[8]*m[2]*m[5];
  double det = m[0]*inv[0] + m[1]*inv[4] + m[2]*inv[8] + m[3]*inv[12];
  if (det == 0.0) return false;
  for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) out.m[c][r] = (float)(inv[c * 4 + r] / det);
  return true;
}

Mat4 transpose(const Mat4& a) {
  Mat4 r;
  for (int c = 0; c < 4; ++c) for (int k = 0; k < 4; ++k) r.m[c][k] = a.m[k][c];
  return r;
}

f3 get3(const Value& v) {  // parsescene.cpp:34-44
  float x[3] = {0, 0, 0};
  for (size_t i = 0; i < v.arr.size() && i < 3; ++i) x[i] = (float)v.arr[i]->GetDouble();
  return f3{x[0], x[1], x[2]};
}
Mat4 get_mat4(const Value& v) {  // parsescene.cpp:21-32: 16 floats, column-major
  Mat4 r = Mat4::identity();
  float x[16];
  memcpy(x, r.m, sizeof(x));
  for (size_t i = 0; i < v.arr.size() && i < 16; ++i) x[i] = (float)v.arr[i]->GetDouble();
  memcpy(r.m, x, sizeof(x));
  return r;
}
f3 opt3(const Value& o, const char* k, f3 d) { return o.HasMember(k) ? get3(o[k]) : d; }
float optf(const Value& o, const char* k, float d) { return o.HasMember(k) ? (float)o[k].GetDouble() : d; }
int opti(const Value& o, const char* k, int d) { return o.HasMember(k) ? o[k].GetInt() : d; }
bool optb(const Value& o, const char* k, bool d) { return o.HasMember(k) ? o[k].GetBool() : d; }
std::string opts(const Value& o, const char* k, const char* d) {
  return o.HasMember(k) ? o[k].GetString() : std::string(d);
}

void make_coordinate(f3 n, f3& u, f3& w) {  // common.h:116-129
  if (fabsf(n.x) > fabsf(n.y)) {
    float inv = 1.0f / sqrtf(n.x * n.x + n.z * n.z);
    w = mk3(n.z * inv, 0.0f, -n.x * inv);
  } else {
    float inv = 1.0f / sqrtf(n.y * n.y + n.z * n.z);
    w = mk3(0.0f, n.z * inv, -n.y * inv);
  }
  u = cross3(w, n);
}

f3 gen_tangent(const Vertex& v1, const Vertex& v2, const Vertex& v3) {  // mesh.cpp:93-112
  f2 duv21{v2.uv.x - v1.uv.x, v2.uv.y - v1.uv.y}, duv31{v3.uv.x - v1.uv.x, v3.uv.y - v1.uv.y};
  f3 dp21 = v2.v - v1.v, dp31 = v3.v - v1.v;
  float det = duv21.x * duv31.y - duv21.y * duv31.x;
  if (!(fabsf(det) < 1e-8f)) {
    float invdet = 1.f / det;
    return normalize3((duv31.y * dp21 - duv21.y * dp31) * invdet);
  }
  f3 uu, ww;
  f3 nn = normalize3(cross3(v2.v - v1.v, v3.v - v1.v));
  make_coordinate(nn, uu, ww);
  return uu;
}

// mesh.cpp:29-91 for one mesh: transform, accumulate per-vertex tangents,
// emit triangles.
bool mesh_triangles(const std::string& path, const Mat4& trs, int mat, std::vector<Prim>& out,
                    std::string& err) {
  RawMesh raw;
  if (!load_mesh_file(path, raw, err)) return false;
  Mat4 inv;
  if (!inverse(trs, inv)) { err = "singular transform for \"" + path + "\""; return false; }
  Mat4 invT = transpose(inv);
  std::vector<Vertex> verts(raw.pos.size());
  for (size_t i = 0; i < raw.pos.size(); ++i) {
    float p[4] = {raw.pos[i].x, raw.pos[i].y, raw.pos[i].z, 1.f}, q[4];
    xform(trs, p, q);
    float n[4] = {raw.nrm[i].x, raw.nrm[i].y, raw.nrm[i].z, 0.f}, nn[4];
    xform(invT, n, nn);
    verts[i].v = mk3(q[0], q[1], q[2]);
    verts[i].n = normalize3(mk3(nn[0], nn[1], nn[2]));
    verts[i].uv = raw.uv.empty() ? f2{0.f, 0.f} : raw.uv[i];
    verts[i].t = mk3(0, 0, 0);
  }
  size_t nf = raw.idx.size() / 3;
  for (size_t f = 0; f < nf; ++f) {
    int a = raw.idx[3 * f], b = raw.idx[3 * f + 1], c = raw.idx[3 * f + 2];
    f3 t = gen_tangent(verts[a], verts[b], verts[c]);
    verts[a].t = verts[a].t + t; verts[b].t = verts[b].t + t; verts[c].t = verts[c].t + t;
  }
  for (auto& v : verts) v.t = normalize3(v.t);
  for (size_t f = 0; f < nf; ++f) {
    Prim p{};
    p.type = GT_TRIANGLE;
    p.v[0] = verts[raw.idx[3 * f]]; p.v[1] = verts[raw.idx[3 * f + 1]]; p.v[2] = verts[raw.idx[3 * f + 2]];
    p.mat = mat; p.light = -1; p.med_in = -1; p.med_out = -1;
    out.push_back(p);
  }
  return true;
}

float tri_area(const Prim& p) {  // Triangle.h:81-85
  f3 e1 = p.v[1].v - p.v[0].v, e2 = p.v[2].v - p.v[0].v;
  f3 c = cross3(e1, e2);
  return sqrtf(dot3(c, c)) * 0.5f;
}

const float kPI = 3.14159265358f;       // common.h:17-19
const float kFOURPI = 12.56637061432f;

bool load_impl(const char* filename, Scene& scene, std::string& err, BvhBuilder builder, void* builder_ctx) {
  const auto t_start = std::chrono::steady_clock::now();
  std::string file = filename;
  std::string base = file.substr(0, file.find_last_of('/') + 1);
  std::ifstream in(filename, std::ios::binary);
  if (!in.good()) { err = std::string("Scene file [\"") + filename + "\"] is not good"; return false; }
  std::ostringstream ss;
  ss << in.rdbuf();
  cjson::ValuePtr root = cjson::parse(ss.str());
  const Value& doc = *root;
  if (!doc.IsObject()) { err = "scene file must hold a JSON object"; return false; }

  // media (parsescene.cpp:74-135)
  std::vector<std::string> medium_name;
  if (doc.HasMember("medium")) {
    const Value& s = doc["medium"];
    if (!s.IsArray()) { err = "Invalid Medium format"; return false; }
    for (auto& it : s.arr) {
      const Value& u = *it;
      std::string type = opts(u, "type", "homogeneous");
      f3 sa = opt3(u, "sigmaA", mk3(1, 1, 1)), sS = opt3(u, "sigmaS", mk3(1, 1, 1));
      float sc = optf(u, "scale", 1.f);
      sa = sa * sc; sS = sS * sc;
      Medium m{};
      m.type = MED_HOMOGENEOUS;
      m.g = optf(u, "g", 0.f);
      m.sigmaA = sa; m.sigmaS = sS; m.sigmaT = sa + sS;
      if (type != "homogeneous") {  // parsescene.cpp:101-131: anything else is the density grid
        if (m.sigmaT.x != m.sigmaT.y || m.sigmaT.x != m.sigmaT.z) {
          err = "sigmaA and sigmaS requires uniform attenuation coefficient";
          return false;
        }
        for (const char* key : {"nx", "ny", "nz", "p0", "p1", "density"})
          if (!u.HasMember(key)) { err = std::string("heterogeneous medium needs \"") + key + "\""; return false; }
        m.type = MED_HETEROGENEOUS;
        m.nx = u["nx"].GetInt(); m.ny = u["ny"].GetInt(); m.nz = u["nz"].GetInt();
        if (m.nx <= 0 || m.ny <= 0 || m.nz <= 0 || (long long)m.nx * m.ny * m.nz > (1ll << 30)) {
          err = "heterogeneous medium: bad grid size";
          return false;
        }
        m.p0 = opt3(u, "p0", mk3(0, 0, 0)); m.p1 = opt3(u, "p1", mk3(1, 1, 1));
        m.iterMax = u.HasMember("iterMax") ? u["iterMax"].GetInt() : 1000;
        // ReadDensityFromFile (medium.h:189-197): whitespace-separated decimal floats
        std::string file = base + u["density"].GetString();
        FILE* fp = fopen(file.c_str(), "r");
        if (!fp) { err = "cannot open density file " + file; return false; }
        m.density.resize((size_t)m.nx * m.ny * m.nz);
        for (float& v : m.density)
          if (fscanf(fp, "%f", &v) != 1) { fclose(fp); err = "density file " + file + " is too short"; return false; }
        fclose(fp);
        float mx = 0.f;
        for (float v : m.density) if (v > mx) mx = v;
        m.invMaxDensity = 1.f / mx;
        scene.has_heterogeneous = true;
      }
      scene.media.push_back(m);
      medium_name.push_back(u["name"].GetString());
    }
  }
  auto get_medium = [&](const std::string& m) {
    for (size_t i = 0; i < medium_name.size(); ++i) if (medium_name[i] == m) return (int)i;
    return -1;
  };

  // globals + camera (parsescene.cpp:149-180)
  if (doc.HasMember("screen_width") && doc.HasMember("screen_height")) {
    scene.width = doc["screen_width"].GetInt();
    scene.height = doc["screen_height"].GetInt();
  } else {
    scene.width = 512; scene.height = 512;
  }
  if (scene.width <= 0 || scene.height <= 0) { err = "bad screen size"; return false; }
  scene.epsilon = optf(doc, "epsilon", 0.001f);
  if (!doc.HasMember("camera")) { err = "Scene file must define camera"; return false; }
  {
    const Value& c = doc["camera"];
    Camera& cam = scene.camera;
    cam.environment = optb(c, "environment", false);
    cam.position = opt3(c, "position", mk3(0, 0, 0));
    cam.fov = optf(c, "fov", 60.f);
    f3 up = opt3(c, "up", mk3(0, 1, 0));
    f3 lookat = opt3(c, "lookat", mk3(0, 0, -1));
    cam.w = normalize3(cam.position - lookat);  // Camera.h:128-134
    cam.u = normalize3(cross3(up, cam.w));
    cam.v = normalize3(cross3(cam.w, cam.u));
    cam.apertureRadius = optf(c, "apertureRadius", 0.f);
    cam.focalDistance = optf(c, "focalDistance", 0.f);
    cam.filmic = optb(c, "filmicTonemap", true);
    cam.medium = get_medium(opts(c, "medium", ""));
    // parsescene.cpp:608 + Camera.h:31-43, distance fixed at 0.1
    cam.res_x = (float)scene.width; cam.res_y = (float)scene.height;
    cam.distance = 0.1f;
    float half_fov = cam.fov * .5f;
    float radians = (float)((double)half_fov / 180.0 * (double)kPI);
    cam.half_h = tanf(radians) * cam.distance;
    cam.half_w = cam.half_h * cam.res_x / cam.res_y;
    cam.area = 4.f * cam.half_w * cam.half_h;
    cam.p2s_x = 2.f * cam.half_w / cam.res_x;
    cam.p2s_y = 2.f * cam.half_h / cam.res_y;
    cam.ratio = cam.focalDistance / cam.distance;
  }

  // integrator (parsescene.cpp:183-225)
  {
    std::string integ = opts(doc, "integrator", "pt");
    if (integ == "ao") { scene.integrator = IT_AO; scene.maxDist = optf(doc, "maxDist", 0.5f); }
    else if (integ == "pt") { scene.integrator = IT_PT; scene.maxDepth = opti(doc, "maxDepth", 5); }
    else if (integ == "vpt") { scene.integrator = IT_VPT; scene.maxDepth = opti(doc, "maxDepth", 5); }
    else if (integ == "lt") { scene.integrator = IT_LT; scene.maxDepth = opti(doc, "maxDepth", 5); }
    else if (integ == "bdpt") { scene.integrator = IT_BDPT; scene.maxDepth = opti(doc, "maxDepth", 5); }
    else {
      err = "Unsupport integrator [" + integ + "]: this renderer implements ao, pt, vpt, lt and bdpt";
      return false;
    }
  }

  // materials (parsescene.cpp:230-329)
  std::vector<std::string> mat_name;
  if (doc.HasMember("material")) {
    const Value& mats = doc["material"];
    if (!mats.IsArray()) { err = "Invalid material format"; return false; }
    std::map<std::string, int> mat_map = {
        {"lambertian", MT_LAMBERTIAN}, {"mirror", MT_MIRROR}, {"dielectric", MT_DIELECTRIC},
        {"roughdielectric", MT_ROUGHDIELECTRIC}, {"roughconduct", MT_ROUGHCONDUCTOR},
        {"substrate", MT_SUBSTRATE}};
    std::map<std::string, int> tex_map;
    for (auto& it : mats.arr) {
      const Value& u = *it;
      if (u.HasMember("bssrdf")) { err = "bssrdf materials are not supported"; return false; }
      std::string bsdf = u["bsdf"].GetString();
      if (!mat_map.count(bsdf)) { err = "unknown bsdf \"" + bsdf + "\""; return false; }
      float aU, aV;
      if (u.HasMember("alpha")) { aU = (float)u["alpha"].GetDouble(); aV = aU; }
      else { aU = optf(u, "alphaU", 0.01f); aV = optf(u, "alphaV", 0.01f); }
      if (optb(u, "remap", false)) {
        auto remap = [](float r) {
          r = r > 1e-3f ? r : 1e-3f;
          float x = logf(r);
          return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x +
                 0.000640711f * x * x * x * x;
        };
        aU = remap(aU); aV = remap(aV);
      }
      Material m{};
      m.type = mat_map[bsdf];
      m.alphaU = aU; m.alphaV = aV;
      m.insideIOR = optf(u, "insideIOR", 1.f);
      m.outsideIOR = optf(u, "outsideIOR", 1.f);
      m.k = opt3(u, "k", mk3(0, 0, 0));
      m.eta = opt3(u, "eta", mk3(0, 0, 0));
      m.specular = opt3(u, "specular", mk3(1, 1, 1));
      m.diffuse = mk3(1, 1, 1);
      m.tex = -1;
      if (u.HasMember("diffuse")) {
        if (u["diffuse"].IsString()) {
          std::string tf = u["diffuse"].GetString();
          if (!tex_map.count(tf)) {
            Texture t;
            if (!load_texture_file(base + tf, t, err)) return false;
            scene.textures.push_back(std::move(t));
            tex_map[tf] = (int)scene.textures.size() - 1;
          }
          m.tex = tex_map[tf];
        } else {
          m.diffuse = get3(u["diffuse"]);
        }
      }
      scene.materials.push_back(m);
      mat_name.push_back(u["name"].GetString());
    }
  }
  auto find_mat = [&](const std::string& n) {
    for (size_t i = 0; i < mat_name.size(); ++i) if (mat_name[i] == n) return (int)i;
    return -1;
  };

  // shapes (parsescene.cpp:332-497)
  std::vector<Prim> prims;
  if (doc.HasMember("scene")) {
    for (auto& it : doc["scene"].arr) {
      const Value& u = *it;
      if (u.HasMember("mesh")) {
        std::string mfile = u["mesh"].GetString();
        std::string mname = opts(u, "material", "");
        Mat4 trs = u.HasMember("transform")
                       ? get_mat4(u["transform"])
                       : trs_of(opt3(u, "translate", mk3(0, 0, 0)), opt3(u, "rotate", mk3(0, 0, 0)),
                                opt3(u, "scale", mk3(1, 1, 1)));
        int mi = get_medium(opts(u, "inside", "")), mo = get_medium(opts(u, "outside", ""));
        int mat = -1;
        if (mname != "" || !(mi != -1 || mo != -1)) {
          mat = find_mat(mname);
          if (mat < 0) { err = "There is no material named:[\"" + mname + "\"]"; return false; }
        }
        size_t first = prims.size();
        if (!mesh_triangles(base + mfile, trs, mat, prims, err)) return false;
        for (size_t i = first; i < prims.size(); ++i) { prims[i].med_in = mi; prims[i].med_out = mo; }
      } else if (u.HasMember("line")) {
        std::string mname = opts(u, "material", "matte");
        Mat4 trs = trs_of(opt3(u, "translate", mk3(0, 0, 0)), opt3(u, "rotate", mk3(0, 0, 0)),
                          opt3(u, "scale", mk3(1, 1, 1)));
        f3 p0 = opt3(u, "p0", mk3(0, 0, 0)), p1 = opt3(u, "p1", mk3(1, 1, 1));
        float a[4] = {p0.x, p0.y, p0.z, 1.f}, b[4] = {p1.x, p1.y, p1.z, 1.f}, qa[4], qb[4];
        xform(trs, a, qa); xform(trs, b, qb);
        Prim p{};
        p.type = GT_LINES;
        p.p0 = mk3(qa[0], qa[1], qa[2]); p.p1 = mk3(qb[0], qb[1], qb[2]);
        p.w0 = optf(u, "width0", 0.025f); p.w1 = optf(u, "width1", 0.025f);
        p.mat = find_mat(mname);
        if (p.mat < 0) { err = "There is no material named:[\"" + mname + "\"]"; return false; }
        p.light = -1; p.med_in = -1; p.med_out = -1;
        prims.push_back(p);
      } else if (u.HasMember("sphere")) {
        std::string mname = opts(u, "material", "");
        int mi = get_medium(opts(u, "inside", "")), mo = get_medium(opts(u, "outside", ""));
        Prim p{};
        p.type = GT_SPHERE;
        p.p0 = opt3(u, "center", mk3(0, 0, 0));
        p.w0 = optf(u, "radius", 1.f);
        p.mat = -1;
        if (mname != "" || !(mi != -1 || mo != -1)) {
          p.mat = find_mat(mname);
          if (p.mat < 0) { err = "There is no material named:[\"" + mname + "\"]"; return false; }
        }
        p.light = -1; p.med_in = mi; p.med_out = mo;
        prims.push_back(p);
      } else {
        err = "Error scene file format";
        return false;
      }
    }
  }

  // lights (parsescene.cpp:499-594)
  std::vector<f3> light_radiance;
  std::vector<int> light_medium;
  if (doc.HasMember("light")) {
    for (auto& it : doc["light"].arr) {
      const Value& u = *it;
      if (u.HasMember("mesh")) {
        std::string mfile = u["mesh"].GetString();
        std::string mname = opts(u, "material", "matte");
        f3 radiance = opt3(u, "radiance", mk3(0, 0, 0));
        int med = get_medium(opts(u, "medium", ""));
        Mat4 trs = trs_of(opt3(u, "translate", mk3(0, 0, 0)), opt3(u, "rotate", mk3(0, 0, 0)),
                          opt3(u, "scale", mk3(1, 1, 1)));
        int mat = find_mat(mname);
        if (mat < 0) { err = "There is no material named:[\"" + mname + "\"]"; return false; }
        size_t first = prims.size();
        if (!mesh_triangles(base + mfile, trs, mat, prims, err)) return false;
        for (size_t i = first; i < prims.size(); ++i) {
          prims[i].light = (int)light_radiance.size();
          light_radiance.push_back(radiance);
          light_medium.push_back(med);
        }
      } else if (u.HasMember("infinite")) {
        Infinite inf;
        if (!load_envmap_file(base + u["infinite"].GetString(), inf, err)) return false;
        Mat4 rs = Mat4::identity();
        bool have = false;
        if (u.HasMember("rotate")) {
          f3 r = get3(u["rotate"]);
          rs = mul(mul(rotate_axis(r.x, 0), rotate_axis(r.y, 1)), rotate_axis(r.z, 2));
          have = true;
        }
        if (u.HasMember("matrix")) {
          Mat4 m = get_mat4(u["matrix"]);
          if (!inverse(m, rs)) { err = "singular environment matrix"; return false; }
          have = true;
        }
        if (have) {
          float ex[4] = {1, 0, 0, 0}, ey[4] = {0, 1, 0, 0}, ez[4] = {0, 0, 1, 0}, o[4];
          xform(rs, ex, o); inf.u = mk3(o[0], o[1], o[2]);
          xform(rs, ey, o); inf.v = mk3(o[0], o[1], o[2]);
          xform(rs, ez, o); inf.w_ = mk3(o[0], o[1], o[2]);
        }
        scene.infinite = std::move(inf);
      } else {
        err = "Only support area and infinite light";
        return false;
      }
    }
  }
  for (size_t i = 0; i < prims.size(); ++i) prims[i].push_id = (int)i;

  // Scene::Init (scene.h:48-81)
  const bool timing = getenv("CASTER_TIMING") != nullptr;
  auto t_build = std::chrono::steady_clock::now();
  if (timing) fprintf(stderr, "caster: scene parsed and meshes loaded in %.3f s (%zu primitives)\n",
                      std::chrono::duration<double>(t_build - t_start).count(), prims.size());
  if (builder) {
    std::vector<BBox> boxes;
    std::vector<int> order;
    prim_boxes(prims, boxes, scene.root_box);
    if (!builder(builder_ctx, boxes, scene.root_box, scene.nodes, order, err)) return false;
    apply_prim_order(prims, order);
  } else {
    build_bvh(prims, scene.nodes, scene.root_box);
  }
  if (timing) fprintf(stderr, "caster: BVH built in %.3f s (%zu nodes)\n",
                      std::chrono::duration<double>(std::chrono::steady_clock::now() - t_build).count(),
                      scene.nodes.size());
  scene.prims.swap(prims);
  if (!scene.prims.empty()) {
    f3 diag = scene.root_box.hi - scene.root_box.lo;
    scene.far_t = kFarFactor * sqrtf(dot3(diag, diag));
  }
  scene.lights.resize(light_radiance.size());
  for (size_t i = 0; i < scene.prims.size(); ++i) {
    int l = scene.prims[i].light;
    if (l >= 0) scene.lights[l] = Light{light_radiance[l], (int)i, light_medium[l]};
  }
  if (scene.infinite.valid) {  // bbox.h:101-104
    scene.infinite.centre = scene.root_box.centre();
    f3 d = scene.root_box.hi - scene.infinite.centre;
    scene.infinite.radius = sqrtf(dot3(d, d));
  }
  f3 luma = mk3(0.212671f, 0.715160f, 0.072169f);
  float sum = 0.f;
  scene.light_cdf.push_back(0.f);
  for (auto& l : scene.lights) {
    f3 power = l.radiance * tri_area(scene.prims[l.prim]) * kPI;  // area.h:32-34
    sum += dot3(luma, power);
    scene.light_cdf.push_back(sum);
  }
  if (scene.infinite.valid) {  // infinite.h:42-44
    f3 power = (kFOURPI * scene.infinite.radius * scene.infinite.radius) * scene.infinite.data[0];
    sum += dot3(luma, power);
    scene.light_cdf.push_back(sum);
  }
  for (auto& c : scene.light_cdf) c /= sum;
  return true;
}

}  // namespace

bool load_scene(const char* path, Scene& out, std::string& err, BvhBuilder builder, void* builder_ctx) {
  try {
    return load_impl(path, out, err, builder, builder_ctx);
  } catch (const std::exception& e) {
    err = e.what();
    return false;
  }
}

}  // namespace caster
