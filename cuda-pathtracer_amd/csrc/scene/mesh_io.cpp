// Asset readers for the scene loader.  The reference goes through assimp, stb
// and tinyexr (CUDA-PathTracer/tracer/mesh.cpp:4-47, imageio.cpp:12-103);
// none of those is available here, so these are small readers for the
// formats the synthetic scenes use: Wavefront OBJ and binary PLY meshes,
// binary PPM / 8-bit PNG textures, PFM environment maps.
#include <zlib.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <unordered_map>
#include <cmath>
#include "scene.h"

namespace caster {

namespace {

bool ends_with(const std::string& s, const char* suf) {
  size_t n = strlen(suf);
  if (s.size() < n) return false;
  for (size_t i = 0; i < n; ++i)
    if (tolower((unsigned char)s[s.size() - n + i]) != suf[i]) return false;
  return true;
}

bool read_file(const std::string& path, std::string& out) {
  std::ifstream in(path, std::ios::binary);
  if (!in.good()) return false;
  std::ostringstream ss;
  ss << in.rdbuf();
  out = ss.str();
  return true;
}

// OBJ: one output vertex per face corner (assimp's OBJ importer does not
// share vertices unless asked to join them), polygons fan-triangulated.
// A file without any normals gets smooth ones (below).
void smooth_normals(RawMesh& m);
bool load_obj(const std::string& path, RawMesh& m, std::string& err) {
  std::string text;
  if (!read_file(path, text)) { err = "cannot open mesh \"" + path + "\""; return false; }
  std::vector<f3> P, N;
  std::vector<f2> T;
  bool any_uv = false;
  const char* s = text.c_str();
  const char* end = s + text.size();
  std::vector<int> corner;  // v, vt, vn triples of the current face
  size_t with_normal = 0, without_normal = 0;
  while (s < end) {
    const char* le = (const char*)memchr(s, '\n', (size_t)(end - s));
    if (!le) le = end;
    while (s < le && (*s == ' ' || *s == '\t')) ++s;
    if (le - s >= 2 && s[0] == 'v' && (s[1] == ' ' || s[1] == '\t')) {
      char* e;
      f3 p;
      p.x = strtof(s + 2, &e); p.y = strtof(e, &e); p.z = strtof(e, &e);
      P.push_back(p);
    } else if (le - s >= 3 && s[0] == 'v' && s[1] == 'n') {
      char* e;
      f3 p;
      p.x = strtof(s + 3, &e); p.y = strtof(e, &e); p.z = strtof(e, &e);
      N.push_back(p);
    } else if (le - s >= 3 && s[0] == 'v' && s[1] == 't') {
      char* e;
      f2 p;
      p.x = strtof(s + 3, &e); p.y = strtof(e, &e);
      T.push_back(p);
    } else if (le - s >= 2 && s[0] == 'f' && (s[1] == ' ' || s[1] == '\t')) {
      corner.clear();
      const char* q = s + 2;
      while (q < le) {
        while (q < le && (*q == ' ' || *q == '\t' || *q == '\r')) ++q;
        if (q >= le) break;
        int v = 0, vt = 0, vn = 0;
        char* e;
        v = (int)strtol(q, &e, 10);
        if (e == q) break;
        q = e;
        if (q < le && *q == '/') {
          ++q;
          if (q < le && *q != '/') { vt = (int)strtol(q, &e, 10); q = e; }
          if (q < le && *q == '/') { ++q; vn = (int)strtol(q, &e, 10); q = e; }
        }
        if (v < 0) v = (int)P.size() + v + 1;
        if (vt < 0) vt = (int)T.size() + vt + 1;
        if (vn < 0) vn = (int)N.size() + vn + 1;
        corner.push_back(v); corner.push_back(vt); corner.push_back(vn);
      }
      int nc = (int)corner.size() / 3;
      for (int k = 1; k + 1 < nc; ++k) {
        int tri[3] = {0, k, k + 1};
        for (int c = 0; c < 3; ++c) {
          int v = corner[3 * tri[c]], vt = corner[3 * tri[c] + 1], vn = corner[3 * tri[c] + 2];
          if (v < 1 || v > (int)P.size()) { err = "bad vertex index in \"" + path + "\""; return false; }
          const bool has_n = vn >= 1 && vn <= (int)N.size();
          ++(has_n ? with_normal : without_normal);
          m.idx.push_back((int)m.pos.size());
          m.pos.push_back(P[v - 1]);
          m.nrm.push_back(has_n ? N[vn - 1] : f3{0.f, 0.f, 0.f});
          if (vt >= 1 && vt <= (int)T.size()) { m.uv.push_back(T[vt - 1]); any_uv = true; }
          else m.uv.push_back(f2{0.f, 0.f});
        }
      }
    }
    s = le + 1;
  }
  if (!any_uv) m.uv.clear();
  if (m.idx.empty()) { err = "mesh \"" + path + "\" has no faces"; return false; }
  if (without_normal && with_normal) {
    err = "mesh \"" + path + "\" has faces with and faces without normals";
    return false;
  }
  if (without_normal) smooth_normals(m);
  return true;
}

// Normals for a mesh that has none, after assimp's aiProcess_GenSmoothNormals, which the reference
// asks for (parsescene.cpp:390,537): every corner takes its triangle's unit normal, and all corners
// at one position share the normalised sum of theirs (no angle limit, assimp's default).  The
// reference's assimp version is not pinned and only ships as a Windows binary, so this is written
// from the algorithm's description, not checked against it; corners are matched by exact position
// where assimp matches within 1e-5 of the mesh's extent.
void smooth_normals(RawMesh& m) {
  struct Key {
    uint32_t x, y, z;
    bool operator==(const Key& o) const { return x == o.x && y == o.y && z == o.z; }
  };
  struct Hash {
    size_t operator()(const Key& k) const { return (size_t)k.x * 73856093u ^ (size_t)k.y * 19349663u ^ (size_t)k.z * 83492791u; }
  };
  auto key = [](f3 p) {
    Key k;
    float c[3] = {p.x + 0.f, p.y + 0.f, p.z + 0.f};  // -0 -> +0
    memcpy(&k, c, 12);
    return k;
  };
  std::unordered_map<Key, f3, Hash> sum;
  std::vector<f3> face(m.idx.size() / 3);
  for (size_t t = 0; t < face.size(); ++t) {
    f3 a = m.pos[(size_t)m.idx[3 * t]], b = m.pos[(size_t)m.idx[3 * t + 1]], c = m.pos[(size_t)m.idx[3 * t + 2]];
    f3 e1 = {b.x - a.x, b.y - a.y, b.z - a.z}, e2 = {c.x - a.x, c.y - a.y, c.z - a.z};
    f3 n = {e1.y * e2.z - e1.z * e2.y, e1.z * e2.x - e1.x * e2.z, e1.x * e2.y - e1.y * e2.x};
    float len = sqrtf(n.x * n.x + n.y * n.y + n.z * n.z);
    if (len > 0.f) n = f3{n.x / len, n.y / len, n.z / len};
    face[t] = n;
    for (int k = 0; k < 3; ++k) {
      f3& s = sum[key(m.pos[(size_t)m.idx[3 * t + k]])];
      s = f3{s.x + n.x, s.y + n.y, s.z + n.z};
    }
  }
  for (size_t t = 0; t < face.size(); ++t)
    for (int k = 0; k < 3; ++k) {
      size_t v = (size_t)m.idx[3 * t + k];
      f3 s = sum[key(m.pos[v])];
      float len = sqrtf(s.x * s.x + s.y * s.y + s.z * s.z);
      m.nrm[v] = len > 0.f ? f3{s.x / len, s.y / len, s.z / len} : face[t];
    }
}

// Binary little-endian PLY with float x,y,z,nx,ny,nz[,s,t|u,v] vertices and
// `list uchar int|uint vertex_indices` faces.  Vertices stay shared.
bool load_ply(const std::string& path, RawMesh& m, std::string& err) {
  std::string data;
  if (!read_file(path, data)) { err = "cannot open mesh \"" + path + "\""; return false; }
  size_t hdr_end = data.find("end_header");
  if (data.compare(0, 3, "ply") != 0 || hdr_end == std::string::npos) {
    err = "\"" + path + "\" is not a PLY file"; return false;
  }
  size_t body = data.find('\n', hdr_end) + 1;
  std::istringstream hs(data.substr(0, hdr_end));
  std::string line;
  int nv = 0, nf = 0, section = 0;
  std::vector<std::string> vprops;
  bool binary = false;
  while (std::getline(hs, line)) {
    std::istringstream ls(line);
    std::string w;
    ls >> w;
    if (w == "format") { ls >> w; binary = (w == "binary_little_endian"); }
    else if (w == "element") {
      std::string name; int cnt = -1;
      ls >> name >> cnt;
      if (ls.fail() || cnt < 0) { err = "PLY \"" + path + "\": bad element count"; return false; }
      if (name == "vertex") { nv = cnt; section = 1; }
      else if (name == "face") { nf = cnt; section = 2; }
      else section = 3;
    } else if (w == "property" && section == 1) {
      std::string ty, name;
      ls >> ty >> name;
      if (ty != "float" && ty != "float32") { err = "PLY vertex properties must be float"; return false; }
      vprops.push_back(name);
    }
  }
  if (!binary) { err = "only binary_little_endian PLY is supported"; return false; }
  auto find = [&](const char* a, const char* b) {
    for (size_t i = 0; i < vprops.size(); ++i) if (vprops[i] == a || vprops[i] == b) return (int)i;
    return -1;
  };
  int ix = find("x", "x"), iy = find("y", "y"), iz = find("z", "z");
  int inx = find("nx", "nx"), iny = find("ny", "ny"), inz = find("nz", "nz");
  int iu = find("s", "u"), iv = find("t", "v");
  if (ix < 0 || iy < 0 || iz < 0 || inx < 0 || iny < 0 || inz < 0) {
    err = "PLY \"" + path + "\" needs x,y,z,nx,ny,nz"; return false;
  }
  if (vprops.size() > 32) { err = "PLY \"" + path + "\" has more than 32 vertex properties"; return false; }
  size_t stride = vprops.size() * 4;
  if (body > data.size() || stride * (size_t)nv > data.size() - body) { err = "truncated PLY"; return false; }
  const char* p = data.data() + body;
  m.pos.resize(nv); m.nrm.resize(nv);
  if (iu >= 0 && iv >= 0) m.uv.resize(nv);
  for (int i = 0; i < nv; ++i) {
    const float* f = (const float*)(p + stride * (size_t)i);
    float tmp[32];
    memcpy(tmp, f, stride > sizeof(tmp) ? sizeof(tmp) : stride);
    m.pos[i] = f3{tmp[ix], tmp[iy], tmp[iz]};
    m.nrm[i] = f3{tmp[inx], tmp[iny], tmp[inz]};
    if (!m.uv.empty()) m.uv[i] = f2{tmp[iu], tmp[iv]};
  }
  const unsigned char* q = (const unsigned char*)p + stride * (size_t)nv;
  const unsigned char* qe = (const unsigned char*)data.data() + data.size();
  m.idx.reserve((size_t)nf * 3);
  for (int i = 0; i < nf; ++i) {
    if (q >= qe) { err = "truncated PLY faces"; return false; }
    int c = *q++;
    if (q + 4 * c > qe) { err = "truncated PLY faces"; return false; }
    int32_t id[16];
    if (c > 16) { err = "PLY polygon too large"; return false; }
    memcpy(id, q, 4 * (size_t)c);
    q += 4 * c;
    for (int k = 1; k + 1 < c; ++k) {
      int tri[3] = {id[0], id[k], id[k + 1]};
      for (int t : tri) {
        if (t < 0 || t >= nv) { err = "bad PLY index"; return false; }
        m.idx.push_back(t);
      }
    }
  }
  if (m.idx.empty()) { err = "mesh \"" + path + "\" has no faces"; return false; }
  return true;
}

// ---- images ----------------------------------------------------------------

// Decoded 8-bit image, top row first, `comp` interleaved channels.
struct Image8 { int w = 0, h = 0, comp = 0; std::vector<uint8_t> px; };

bool load_ppm(const std::string& data, Image8& im) {
  if (data.size() < 2 || data[0] != 'P' || (data[1] != '6' && data[1] != '5')) return false;
  im.comp = data[1] == '6' ? 3 : 1;
  size_t p = 2;
  int vals[3], got = 0;
  while (got < 3 && p < data.size()) {
    while (p < data.size() && isspace((unsigned char)data[p])) ++p;
    if (p < data.size() && data[p] == '#') { while (p < data.size() && data[p] != '\n') ++p; continue; }
    int v = 0; bool any = false;
    while (p < data.size() && isdigit((unsigned char)data[p])) { v = v * 10 + (data[p++] - '0'); any = true; }
    if (!any) return false;
    vals[got++] = v;
  }
  if (got < 3 || vals[2] != 255) return false;
  ++p;  // single whitespace after maxval
  im.w = vals[0]; im.h = vals[1];
  size_t n = (size_t)im.w * im.h * im.comp;
  if (p + n > data.size()) return false;
  im.px.assign(data.begin() + p, data.begin() + p + n);
  return true;
}

// Huffman-coded JPEG, baseline, extended sequential or progressive, 8 bits, one or three
// components, sampling factors 1 or 2, interleaved or one scan per component, restart intervals:
// what the reference gets from stb_image for such a file (imageio.cpp:12-17) up to the last bit
// of the inverse DCT and the colour conversion, which differ between decoders (stb's are
// fixed-point; these are floating-point): parity unpinned to +-1 level.  Arithmetic-coded,
// lossless, 12-bit and CMYK files are refused.
struct JpegHuff {
  uint8_t bits[17] = {0};
  uint8_t vals[256] = {0};
  int mincode[17] = {0}, maxcode[18] = {0}, valptr[17] = {0};
  bool set = false;
  void build() {
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
      valptr[l] = k;
      mincode[l] = code;
      code += bits[l];
      k += bits[l];
      maxcode[l] = bits[l] ? code - 1 : -1;
      code <<= 1;
    }
    set = true;
  }
};
struct JpegBits {
  const uint8_t* p; size_t n, at;
  uint32_t acc = 0; int cnt = 0; bool hit_marker = false;
  int bit() {
    if (cnt == 0) {
      uint8_t b = 0;
      if (!hit_marker && at < n) {
        b = p[at++];
        if (b == 0xff) {
          uint8_t b2 = at < n ? p[at] : 0xd9;
          if (b2 == 0) ++at;                       // a stuffed zero after a data byte 0xff
          else { hit_marker = true; --at; b = 0; }  // a marker: the scan's data end here
        }
      }
      acc = b; cnt = 8;
    }
    --cnt;
    return (acc >> cnt) & 1;
  }
  int receive(int s) { int v = 0; for (int i = 0; i < s; ++i) v = (v << 1) | bit(); return v; }
  void reset() { cnt = 0; hit_marker = false; }
};
inline int jpeg_extend(int v, int s) { return s && v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }
inline int jpeg_decode(JpegBits& br, const JpegHuff& h) {
  int code = 0;
  for (int l = 1; l <= 16; ++l) {
    code = (code << 1) | br.bit();
    if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
  }
  return -1;
}
bool load_jpeg(const std::string& data, Image8& im, std::string& err) {
  static const uint8_t zigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                                     41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                                     15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
  const uint8_t* p = (const uint8_t*)data.data();
  const size_t n = data.size();
  if (n < 4 || p[0] != 0xff || p[1] != 0xd8) return false;
  uint16_t qt[4][64] = {{0}};
  JpegHuff hdc[4], hac[4];
  // coef: the quantised coefficients of every block of the component, in zigzag order, filled
  // scan by scan (one scan for a baseline file, many for a progressive one)
  struct Comp { int id, h, v, tq, td, ta, pred; int pw, ph, bw, bh; std::vector<int> coef; std::vector<uint8_t> plane; };
  std::vector<Comp> comps;
  int W = 0, H = 0, restart = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0;
  size_t at = 2;
  bool have_frame = false, progressive = false, have_scan = false;
  while (at + 4 <= n) {
    if (p[at] != 0xff) { ++at; continue; }
    uint8_t m = p[at + 1];
    at += 2;
    if (m == 0xd8 || m <= 0x01 || (m >= 0xd0 && m <= 0xd7) || m == 0xff) { if (m == 0xff) --at; continue; }
    if (m == 0xd9) break;
    if (at + 2 > n) break;
    size_t len = ((size_t)p[at] << 8) | p[at + 1];
    if (len < 2 || at + len > n) { err = "truncated JPEG segment"; return false; }
    const uint8_t* q = p + at + 2;
    const uint8_t* qe = p + at + len;
    if (m == 0xdb) {
      while (q < qe) {
        int pq = q[0] >> 4, tq = q[0] & 15;
        ++q;
        if (tq > 3 || q + (pq ? 128 : 64) > qe) { err = "bad JPEG quantisation table"; return false; }
        for (int i = 0; i < 64; ++i) { qt[tq][i] = pq ? (uint16_t)((q[0] << 8) | q[1]) : q[0]; q += pq ? 2 : 1; }
      }
    } else if (m == 0xc4) {
      while (q < qe) {
        int tc = q[0] >> 4, th = q[0] & 15;
        ++q;
        if (tc > 1 || th > 3 || q + 16 > qe) { err = "bad JPEG Huffman table"; return false; }
        JpegHuff& h = tc ? hac[th] : hdc[th];
        int total = 0;
        for (int l = 1; l <= 16; ++l) { h.bits[l] = q[l - 1]; total += q[l - 1]; }
        q += 16;
        if (total > 256 || q + total > qe) { err = "bad JPEG Huffman table"; return false; }
        memcpy(h.vals, q, (size_t)total);
        q += total;
        h.build();
      }
    } else if (m == 0xc0 || m == 0xc1 || m == 0xc2) {
      if (have_frame) { err = "JPEG with two frame headers"; return false; }
      if (qe - q < 6 || q[0] != 8) { err = "only 8-bit JPEG is supported"; return false; }
      progressive = m == 0xc2;
      H = (q[1] << 8) | q[2]; W = (q[3] << 8) | q[4];
      int nc = q[5];
      if ((nc != 1 && nc != 3) || qe - q < 6 + 3 * nc || W <= 0 || H <= 0) { err = "JPEG with 1 or 3 components expected"; return false; }
      if ((size_t)W * H > (1u << 28)) { err = "JPEG too large"; return false; }
      comps.resize((size_t)nc);
      for (int c = 0; c < nc; ++c) {
        comps[c].id = q[6 + 3 * c]; comps[c].h = q[7 + 3 * c] >> 4; comps[c].v = q[7 + 3 * c] & 15; comps[c].tq = q[8 + 3 * c] & 3;
        if (comps[c].h < 1 || comps[c].h > 2 || comps[c].v < 1 || comps[c].v > 2) { err = "JPEG sampling factors above 2 are not supported"; return false; }
      }
      if (nc == 1) comps[0].h = comps[0].v = 1;
      for (auto& c : comps) { hmax = std::max(hmax, c.h); vmax = std::max(vmax, c.v); }
      mcux = (W + 8 * hmax - 1) / (8 * hmax); mcuy = (H + 8 * vmax - 1) / (8 * vmax);
      for (auto& c : comps) {
        c.pw = mcux * 8 * c.h; c.ph = mcuy * 8 * c.v;
        // blocks that hold picture samples: a scan of this component alone codes only these
        c.bw = ((W * c.h + hmax - 1) / hmax + 7) / 8; c.bh = ((H * c.v + vmax - 1) / vmax + 7) / 8;
        c.coef.assign((size_t)c.pw * c.ph, 0);
      }
      have_frame = true;
    } else if (m >= 0xc3 && m <= 0xcf && m != 0xc8 && m != 0xcc) {
      err = "lossless, hierarchical and arithmetic-coded JPEG files are not supported";
      return false;
    } else if (m == 0xdd) {
      if (qe - q >= 2) restart = (q[0] << 8) | q[1];
    } else if (m == 0xda) {
      if (!have_frame) { err = "JPEG scan before its frame header"; return false; }
      const int ns = q[0];
      if (ns < 1 || ns > (int)comps.size() || qe - q < 1 + 2 * ns + 3) { err = "bad JPEG scan header"; return false; }
      Comp* sc[3];
      for (int k = 0; k < ns; ++k) {
        sc[k] = nullptr;
        for (auto& c : comps)
          if (c.id == q[1 + 2 * k]) { c.td = q[2 + 2 * k] >> 4; c.ta = q[2 + 2 * k] & 15; sc[k] = &c; }
        if (!sc[k] || sc[k]->td > 3 || sc[k]->ta > 3) { err = "bad JPEG scan header"; return false; }
      }
      int ss = q[1 + 2 * ns], se = q[2 + 2 * ns], ah = q[3 + 2 * ns] >> 4, al = q[3 + 2 * ns] & 15;
      if (!progressive) { ss = 0; se = 63; ah = al = 0; }
      if (ss > se || se > 63 || al > 13 || ah > 13 || (ss == 0 && se != 0 && progressive) || (ss > 0 && ns != 1)) {
        err = "bad JPEG spectral selection"; return false;
      }
      for (int k = 0; k < ns; ++k) {
        const bool need_dc = ss == 0 && ah == 0, need_ac = se > 0;
        if ((need_dc && !hdc[sc[k]->td].set) || (need_ac && !hac[sc[k]->ta].set)) { err = "JPEG scan without its Huffman table"; return false; }
        sc[k]->pred = 0;
      }
      JpegBits br{p, n, at + len};
      int eobrun = 0;
      bool bad = false;
      // one block of one scan; b points at the block's 64 coefficients
      auto block = [&](Comp& c, int* b) {
        if (!progressive) {
          int t = jpeg_decode(br, hdc[c.td]);
          if (t < 0 || t > 11) { bad = true; return; }
          c.pred += jpeg_extend(br.receive(t), t);
          b[0] = c.pred;
          for (int k = 1; k < 64;) {
            int rs = jpeg_decode(br, hac[c.ta]);
            if (rs < 0) { bad = true; return; }
            int r = rs >> 4, sz = rs & 15;
            if (sz == 0) { if (r == 15) { k += 16; continue; } break; }
            k += r;
            if (k > 63) { bad = true; return; }
            b[k++] = jpeg_extend(br.receive(sz), sz);
          }
        } else if (ss == 0) {
          if (ah == 0) {               // first pass over the DC coefficients
            int t = jpeg_decode(br, hdc[c.td]);
            if (t < 0 || t > 11) { bad = true; return; }
            c.pred += jpeg_extend(br.receive(t), t);
            b[0] = c.pred * (1 << al);
          } else if (br.bit()) b[0] |= 1 << al;   // one more bit of each
        } else if (ah == 0) {          // first pass over a band of AC coefficients
          if (eobrun > 0) { --eobrun; return; }
          for (int k = ss; k <= se;) {
            int rs = jpeg_decode(br, hac[c.ta]);
            if (rs < 0) { bad = true; return; }
            int r = rs >> 4, sz = rs & 15;
            if (sz == 0) {
              if (r < 15) { eobrun = (1 << r) - 1; if (r) eobrun += br.receive(r); break; }
              k += 16;
            } else {
              k += r;
              if (k > se) { bad = true; return; }
              b[k++] = jpeg_extend(br.receive(sz), sz) * (1 << al);
            }
          }
        } else {                       // one more bit of a band of AC coefficients
          const int bit = 1 << al;
          auto refine = [&](int& v) {
            if (br.bit() && (v & bit) == 0) v += v > 0 ? bit : -bit;
          };
          if (eobrun > 0) {
            --eobrun;
            for (int k = ss; k <= se; ++k) if (b[k]) refine(b[k]);
            return;
          }
          for (int k = ss; k <= se;) {
            int rs = jpeg_decode(br, hac[c.ta]);
            if (rs < 0) { bad = true; return; }
            int r = rs >> 4, sz = rs & 15, val = 0;
            if (sz == 0) {
              if (r < 15) { eobrun = (1 << r) - 1; if (r) eobrun += br.receive(r); r = 64; }  // the rest of the block only refines
            } else {
              if (sz != 1) { bad = true; return; }
              val = br.bit() ? bit : -bit;
            }
            while (k <= se) {          // skip r zero coefficients, refining the others on the way
              int& v = b[k++];
              if (v) refine(v);
              else {
                if (r == 0) { v = val; break; }
                --r;
              }
            }
          }
        }
      };
      int until_restart = restart;
      auto at_restart = [&]() {
        if (restart && until_restart == 0) {
          // the restart marker follows the byte-aligned data of the interval
          br.reset();
          while (br.at + 1 < n && !(p[br.at] == 0xff && p[br.at + 1] >= 0xd0 && p[br.at + 1] <= 0xd7)) ++br.at;
          br.at += 2;
          for (int k = 0; k < ns; ++k) sc[k]->pred = 0;
          eobrun = 0;
          until_restart = restart;
        }
        --until_restart;
      };
      if (ns == 1) {                   // one component: its blocks in raster order
        Comp& c = *sc[0];
        const int row_blocks = c.pw / 8;
        for (int by = 0; by < c.bh && !bad; ++by)
          for (int bx = 0; bx < c.bw && !bad; ++bx) {
            at_restart();
            block(c, c.coef.data() + ((size_t)by * row_blocks + bx) * 64);
          }
      } else {
        for (int my = 0; my < mcuy && !bad; ++my)
          for (int mx = 0; mx < mcux && !bad; ++mx) {
            at_restart();
            for (int k = 0; k < ns; ++k) {
              Comp& c = *sc[k];
              for (int by = 0; by < c.v; ++by)
                for (int bx = 0; bx < c.h && !bad; ++bx)
                  block(c, c.coef.data() + ((size_t)(my * c.v + by) * (c.pw / 8) + (mx * c.h + bx)) * 64);
            }
          }
      }
      if (bad) { err = "bad JPEG entropy data"; return false; }
      have_scan = true;
      at = br.at;
      continue;
    }
    at += len;
  }
  if (!have_scan) { err = "JPEG file without image data"; return false; }

  // coefficients -> samples: dequantise, inverse DCT (separable, float), level shift
  float cosv[8][8];
  for (int x = 0; x < 8; ++x)
    for (int u = 0; u < 8; ++u) cosv[x][u] = (u == 0 ? 0.70710678118654752f : 1.f) * cosf((2 * x + 1) * u * 3.14159265358979323846f / 16.f);
  for (auto& c : comps) {
    c.plane.assign((size_t)c.pw * c.ph, 0);
    const int row_blocks = c.pw / 8;
    for (int by = 0; by < c.ph / 8; ++by)
      for (int bx = 0; bx < row_blocks; ++bx) {
        const int* b = c.coef.data() + ((size_t)by * row_blocks + bx) * 64;
        float coef[64], tmp[64];
        for (int k = 0; k < 64; ++k) coef[zigzag[k]] = (float)(b[k] * qt[c.tq][k]);
        for (int y = 0; y < 8; ++y)      // rows
          for (int x = 0; x < 8; ++x) {
            float a = 0.f;
            for (int u = 0; u < 8; ++u) a += cosv[x][u] * coef[y * 8 + u];
            tmp[y * 8 + x] = a * 0.5f;
          }
        uint8_t* dst = c.plane.data() + (size_t)(by * 8) * c.pw + (size_t)bx * 8;
        for (int x = 0; x < 8; ++x)      // columns
          for (int y = 0; y < 8; ++y) {
            float a = 0.f;
            for (int v = 0; v < 8; ++v) a += cosv[y][v] * tmp[v * 8 + x];
            float val = a * 0.5f + 128.f;
            dst[(size_t)y * c.pw + x] = (uint8_t)(val < 0.f ? 0 : val > 255.f ? 255 : (int)(val + 0.5f));
          }
      }
  }
  // chroma up to full resolution with a triangle filter (sample centres of a 2:1 plane sit
  // between the luma samples), then YCbCr -> RGB
  im.w = W; im.h = H; im.comp = (int)comps.size();
  im.px.assign((size_t)W * H * im.comp, 0);
  auto sample = [&](const Comp& c, int x, int y) -> float {
    auto at1 = [](int i, int f, float& w) -> int {   // position of full-res index i in a plane subsampled by f
      if (f == 1) { w = 0.f; return i; }
      float pos = (i + 0.5f) / f - 0.5f;
      int i0 = (int)floorf(pos);
      w = pos - i0;
      return i0;
    };
    int fx = hmax / c.h, fy = vmax / c.v;
    float wx, wy;
    int x0 = at1(x, fx, wx), y0 = at1(y, fy, wy);
    int cw = (W + fx - 1) / fx, chh = (H + fy - 1) / fy;
    auto px = [&](int xx, int yy) -> float {
      xx = xx < 0 ? 0 : xx >= cw ? cw - 1 : xx;
      yy = yy < 0 ? 0 : yy >= chh ? chh - 1 : yy;
      return (float)c.plane[(size_t)yy * c.pw + xx];
    };
    if (fx == 1 && fy == 1) return px(x0, y0);
    return (1 - wy) * ((1 - wx) * px(x0, y0) + wx * px(x0 + 1, y0)) + wy * ((1 - wx) * px(x0, y0 + 1) + wx * px(x0 + 1, y0 + 1));
  };
  auto clamp8 = [](float v) -> uint8_t { return (uint8_t)(v < 0.f ? 0 : v > 255.f ? 255 : (int)(v + 0.5f)); };
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      uint8_t* o = im.px.data() + ((size_t)y * W + x) * im.comp;
      float Y = sample(comps[0], x, y);
      if (im.comp == 1) { o[0] = clamp8(Y); continue; }
      float cb = sample(comps[1], x, y) - 128.f, cr = sample(comps[2], x, y) - 128.f;
      o[0] = clamp8(Y + 1.402f * cr);
      o[1] = clamp8(Y - 0.344136f * cb - 0.714136f * cr);
      o[2] = clamp8(Y + 1.772f * cb);
    }
  return true;
}

// Bounds-checked little-endian cursor over a file image, for the BMP and TGA headers.
struct LeReader {
  const uint8_t* p; size_t n, at = 0; bool over = false;
  uint32_t u8() { if (at >= n) { over = true; return 0; } return p[at++]; }
  uint32_t u16() { uint32_t a = u8(); return a | (u8() << 8); }
  uint32_t u32() { uint32_t a = u16(); return a | (u16() << 16); }
  void skip(size_t k) { at += k; if (at > n) { at = n; over = true; } }
};

// A masked field of a 16/32-bit BMP pixel widened to 8 bits by repeating its bits, as
// stb_image's stbi__shiftsigned does for fields of one to eight bits.
inline int bmp_field(uint32_t v, uint32_t mask) {
  if (!mask) return 0;
  int lo = 0, bits = 0;
  while (!((mask >> lo) & 1)) ++lo;
  while (lo + bits < 32 && ((mask >> (lo + bits)) & 1)) ++bits;
  uint32_t f = (v & mask) >> lo;
  if (bits >= 8) return (int)(f >> (bits - 8));
  uint32_t r = 0;
  int have = 0;
  while (have < 8) { r = (r << bits) | f; have += bits; }
  return (int)(r >> (have - 8));
}

// Windows BMP as stb_image 2.19 reads it for the reference (imageio.cpp:15): 1/4/8-bit
// palettes, 16/24/32-bit pixels with the default or BI_BITFIELDS masks, bottom-up or top-down;
// three components unless an alpha mask exists; a 32-bit BI_RGB file whose fourth bytes are all
// zero comes back opaque.  RLE files are refused, as there.
bool load_bmp(const std::string& data, Image8& im, std::string& err) {
  LeReader r{(const uint8_t*)data.data(), data.size()};
  if (r.u8() != 'B' || r.u8() != 'M') { err = "not a BMP"; return false; }
  r.skip(8);
  const uint32_t offset = r.u32(), hsz = r.u32();
  if (hsz != 12 && hsz != 40 && hsz != 56 && hsz != 108 && hsz != 124) { err = "unknown BMP header"; return false; }
  int w, h;
  if (hsz == 12) { w = (int)r.u16(); h = (int)r.u16(); } else { w = (int)r.u32(); h = (int)r.u32(); }
  if (r.u16() != 1) { err = "bad BMP"; return false; }
  const int bpp = (int)r.u16();
  uint32_t mr = 0, mg = 0, mb = 0, ma = 0, all_a = 255;
  if (hsz != 12) {
    const uint32_t compress = r.u32();
    if (compress == 1 || compress == 2) { err = "run-length encoded BMP is not supported"; return false; }
    r.skip(20);
    if (hsz == 40 || hsz == 56) {
      if (hsz == 56) r.skip(16);
      if (bpp == 16 || bpp == 32) {
        if (compress == 0) {
          if (bpp == 32) { mr = 0xffu << 16; mg = 0xffu << 8; mb = 0xffu; ma = 0xffu << 24; all_a = 0; }
          else { mr = 31u << 10; mg = 31u << 5; mb = 31u; }
        } else if (compress == 3) {
          mr = r.u32(); mg = r.u32(); mb = r.u32();
          if (mr == mg && mg == mb) { err = "bad BMP masks"; return false; }
        } else { err = "bad BMP compression"; return false; }
      }
    } else {
      mr = r.u32(); mg = r.u32(); mb = r.u32(); ma = r.u32();
    }
  }
  const bool bottom_up = h > 0;
  h = h < 0 ? -h : h;
  if (w <= 0 || h <= 0 || (size_t)w * h > (1u << 28) || r.over) { err = "bad BMP header"; return false; }
  im.w = w; im.h = h; im.comp = ma ? 4 : 3;
  im.px.assign((size_t)w * h * im.comp, 255);
  auto out_row = [&](int j) { return im.px.data() + (size_t)(bottom_up ? h - 1 - j : j) * w * im.comp; };
  if (bpp < 16) {
    if (bpp != 1 && bpp != 4 && bpp != 8) { err = "bad BMP bit depth"; return false; }
    if (bpp == 1 && ma) { err = "1-bit BMP with an alpha mask"; return false; }
    const int esz = hsz == 12 ? 3 : 4;
    // (stb_image 2.19 counts four entries too few under a 12-byte header and leaves them
    // uninitialised; here the whole palette is read)
    const long psize = hsz == 12 ? ((long)offset - 14 - 12) / 3 : ((long)offset - 14 - (long)hsz) >> 2;
    if (psize <= 0 || psize > 256) { err = "bad BMP palette"; return false; }
    uint8_t pal[256][3] = {{0}};
    r.at = 14 + hsz;
    for (long i = 0; i < psize; ++i) { pal[i][2] = (uint8_t)r.u8(); pal[i][1] = (uint8_t)r.u8(); pal[i][0] = (uint8_t)r.u8(); if (esz == 4) r.u8(); }
    r.at = offset;
    const int row_bytes = (w * bpp + 7) / 8, pad = (-row_bytes) & 3;
    for (int j = 0; j < h; ++j) {
      uint8_t* o = out_row(j);
      if (r.at + (size_t)row_bytes > r.n) { err = "truncated BMP"; return false; }
      const uint8_t* row = r.p + r.at;
      r.skip((size_t)row_bytes + pad);
      for (int i = 0; i < w; ++i) {
        int v = bpp == 8 ? row[i] : bpp == 4 ? (row[i >> 1] >> ((i & 1) ? 0 : 4)) & 15 : (row[i >> 3] >> (7 - (i & 7))) & 1;
        o[i * im.comp] = pal[v][0]; o[i * im.comp + 1] = pal[v][1]; o[i * im.comp + 2] = pal[v][2];
      }
    }
    return true;
  }
  if (bpp != 16 && bpp != 24 && bpp != 32) { err = "bad BMP bit depth"; return false; }
  const int easy = bpp == 24 ? 1 : (bpp == 32 && mb == 0xff && mg == 0xff00 && mr == 0x00ff0000 && ma == 0xff000000u) ? 2 : 0;
  if (!easy && (!mr || !mg || !mb)) { err = "bad BMP masks"; return false; }
  const int px_bytes = bpp / 8, pad = bpp == 32 ? 0 : (-(w * px_bytes)) & 3;
  r.at = offset;
  if ((size_t)offset + ((size_t)w * px_bytes + pad) * (size_t)h - pad > r.n) { err = "truncated BMP"; return false; }
  for (int j = 0; j < h; ++j) {
    uint8_t* o = out_row(j);
    for (int i = 0; i < w; ++i, o += im.comp) {
      uint32_t a;
      if (easy) {
        o[2] = (uint8_t)r.u8(); o[1] = (uint8_t)r.u8(); o[0] = (uint8_t)r.u8();
        a = easy == 2 ? r.u8() : 255;
      } else {
        const uint32_t v = bpp == 16 ? r.u16() : r.u32();
        o[0] = (uint8_t)bmp_field(v, mr); o[1] = (uint8_t)bmp_field(v, mg); o[2] = (uint8_t)bmp_field(v, mb);
        a = ma ? (uint32_t)bmp_field(v, ma) : 255;
      }
      all_a |= a;
      if (im.comp == 4) o[3] = (uint8_t)a;
    }
    r.skip(pad);
  }
  if (im.comp == 4 && all_a == 0)
    for (size_t i = 3; i < im.px.size(); i += 4) im.px[i] = 255;
  return true;
}

// Truevision TGA as stb_image 2.19 reads it for the reference (imageio.cpp:15): grey, 15/16-bit
// (five bits a channel, no alpha), 24 and 32-bit pixels, raw or run-length packets, direct or
// through a colour map, rows bottom-up unless bit 5 of the descriptor says top-down.  16-bit grey
// (grey+alpha) is refused: the reference has no branch for two components.
bool load_tga(const std::string& data, Image8& im, std::string& err) {
  LeReader r{(const uint8_t*)data.data(), data.size()};
  const int id_len = (int)r.u8(), indexed = (int)r.u8();
  int type = (int)r.u8();
  const int pal_start = (int)r.u16(), pal_len = (int)r.u16(), pal_bits = (int)r.u8();
  r.skip(4);
  const int w = (int)r.u16(), h = (int)r.u16(), bpp = (int)r.u8(), desc = (int)r.u8();
  const bool rle = type >= 8;
  if (rle) type -= 8;
  if (r.over || indexed > 1 || (indexed ? type != 1 : (type != 2 && type != 3)) || w <= 0 || h <= 0 || (size_t)w * h > (1u << 28)) { err = "not a TGA (or an image format that is not supported)"; return false; }
  const int fmt_bits = indexed ? pal_bits : bpp;
  const bool rgb16 = fmt_bits == 15 || (fmt_bits == 16 && !(type == 3 && !indexed));
  const int comp = fmt_bits == 8 ? 1 : rgb16 ? 3 : (fmt_bits == 24 || fmt_bits == 32) ? fmt_bits / 8 : 0;
  if (fmt_bits == 16 && !rgb16) { err = "grey+alpha TGA (the reference reads two-component images as garbage)"; return false; }
  if (!comp || (indexed && bpp != 8 && bpp != 16)) { err = "unsupported TGA pixel format"; return false; }
  auto rgb16_to = [](uint32_t px, uint8_t* o) {
    o[0] = (uint8_t)((((px >> 10) & 31) * 255) / 31); o[1] = (uint8_t)((((px >> 5) & 31) * 255) / 31); o[2] = (uint8_t)(((px & 31) * 255) / 31);
  };
  r.skip((size_t)id_len);
  std::vector<uint8_t> pal;
  if (indexed) {
    r.skip((size_t)pal_start);
    pal.assign((size_t)std::max(pal_len, 1) * comp, 0);
    for (int i = 0; i < pal_len; ++i) {
      if (rgb16) rgb16_to(r.u16(), pal.data() + (size_t)i * comp);
      else for (int j = 0; j < comp; ++j) pal[(size_t)i * comp + j] = (uint8_t)r.u8();
    }
  }
  im.w = w; im.h = h; im.comp = comp;
  im.px.assign((size_t)w * h * comp, 0);
  const bool bottom_up = !((desc >> 5) & 1);
  uint8_t raw[4] = {0, 0, 0, 0};
  int count = 0;
  bool repeating = false, read_next = true;
  for (size_t i = 0; i < (size_t)w * h; ++i) {
    if (rle) {
      if (count == 0) { const uint32_t cmd = r.u8(); count = 1 + (int)(cmd & 127); repeating = cmd >> 7; read_next = true; }
      else if (!repeating) read_next = true;
    } else read_next = true;
    if (read_next) {
      if (indexed) {
        uint32_t idx = bpp == 8 ? r.u8() : r.u16();
        if ((int)idx >= pal_len) idx = 0;
        memcpy(raw, pal.data() + (size_t)idx * comp, (size_t)comp);
      } else if (rgb16) rgb16_to(r.u16(), raw);
      else for (int j = 0; j < comp; ++j) raw[j] = (uint8_t)r.u8();
      read_next = false;
    }
    const size_t y = i / w, x = i % w;
    uint8_t* o = im.px.data() + ((bottom_up ? (size_t)h - 1 - y : y) * w + x) * comp;
    memcpy(o, raw, (size_t)comp);
    if (comp >= 3 && !rgb16) std::swap(o[0], o[2]);   // stored blue first
    --count;
  }
  if (r.over) { err = "truncated TGA"; return false; }
  return true;
}

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }

// Undo the per-row PNG filters of one (sub-)image in place: `rows` rows of
// 1 filter byte + `stride` data bytes each.
bool png_unfilter(uint8_t* raw, size_t stride, int rows, int bpp) {
  for (int y = 0; y < rows; ++y) {
    uint8_t* row = raw + (stride + 1) * y;
    const uint8_t* up = y ? row - stride : nullptr;  // previous row's data starts one byte later
    const int ft = row[0];
    if (ft > 4) return false;
    uint8_t* out = row + 1;
    for (size_t x = 0; x < stride; ++x) {
      int a = x >= (size_t)bpp ? out[x - bpp] : 0;
      int b = up ? up[x] : 0;
      int c = (up && x >= (size_t)bpp) ? up[x - bpp] : 0;
      int v = out[x];
      switch (ft) {
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) / 2; break;
        case 4: {
          int pa = abs(b - c), pb = abs(a - c), pc = abs(a + b - 2 * c);
          v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
          break;
        }
        default: break;
      }
      out[x] = (uint8_t)v;
    }
  }
  return true;
}

// PNG: grey (1-16 bit), RGB, RGBA (8/16 bit) and palette (1-8 bit) images,
// plain or Adam7-interlaced, with the 8-bit results stb_image hands the
// reference (imageio.cpp:15): grey samples below 8 bits are spread over
// 0..255, 16-bit samples keep their high byte, a palette becomes RGB (RGBA
// with a tRNS chunk), an RGB colour key becomes an alpha channel.  Grey+alpha
// is refused: the reference has no branch for two components.
bool load_png(const std::string& data, Image8& im, std::string& err) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (data.size() < 8 || memcmp(data.data(), sig, 8) != 0) { err = "not a PNG"; return false; }
  const uint8_t* d = (const uint8_t*)data.data();
  size_t p = 8;
  std::vector<uint8_t> z;
  int depth = 0, ctype = 0, interlace = 0, n_pal = 0, n_trns = 0;
  uint8_t pal[256 * 4];
  uint8_t trns[256];
  for (int i = 0; i < 256; ++i) pal[4 * i + 3] = 255;
  bool have_trns = false;
  while (p + 8 <= data.size()) {
    uint32_t len = be32(d + p);
    std::string ty(data, p + 4, 4);
    if (p + 12 + (size_t)len > data.size()) break;
    const uint8_t* body = d + p + 8;
    if (ty == "IHDR" && len >= 13) {
      im.w = (int)be32(body); im.h = (int)be32(body + 4);
      depth = body[8]; ctype = body[9]; interlace = body[12];
    } else if (ty == "PLTE") {
      n_pal = (int)std::min<uint32_t>(len / 3, 256);
      for (int i = 0; i < n_pal; ++i) { pal[4 * i] = body[3 * i]; pal[4 * i + 1] = body[3 * i + 1]; pal[4 * i + 2] = body[3 * i + 2]; }
    } else if (ty == "tRNS") {
      have_trns = true;
      n_trns = (int)std::min<uint32_t>(len, 256);
      memcpy(trns, body, n_trns);
    } else if (ty == "IDAT") {
      z.insert(z.end(), body, body + len);
    } else if (ty == "IEND") break;
    p += 12 + len;
  }
  int chan = 0;
  bool ok_depth = false;
  switch (ctype) {
    case 0: chan = 1; ok_depth = depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16; break;
    case 2: chan = 3; ok_depth = depth == 8 || depth == 16; break;
    case 3: chan = 1; ok_depth = depth == 1 || depth == 2 || depth == 4 || depth == 8; break;
    case 6: chan = 4; ok_depth = depth == 8 || depth == 16; break;
    default: break;
  }
  if (im.w <= 0 || im.h <= 0 || (size_t)im.w * im.h > (1u << 28)) { err = "bad PNG header"; return false; }
  if (ctype == 4 || (ctype == 0 && have_trns)) { err = "grey+alpha PNG (the reference reads two-component images as garbage)"; return false; }
  if (!ok_depth || interlace > 1) { err = "unsupported PNG colour type / bit depth"; return false; }
  if (ctype == 3 && n_pal == 0) { err = "palette PNG without PLTE"; return false; }
  if (ctype == 2 && have_trns && n_trns < 6) { err = "short tRNS chunk"; return false; }

  // Sub-images: the whole picture, or the seven Adam7 passes.
  static const int x0[7] = {0, 4, 0, 2, 0, 1, 0}, y0[7] = {0, 0, 4, 0, 2, 0, 1};
  static const int dx[7] = {8, 8, 4, 4, 2, 2, 1}, dy[7] = {8, 8, 8, 4, 4, 2, 2};
  struct Pass { int w, h, x0, y0, dx, dy; size_t stride, off; };
  std::vector<Pass> passes;
  size_t total = 0;
  const int bits = chan * depth;
  for (int i = 0; i < (interlace ? 7 : 1); ++i) {
    Pass q;
    if (interlace) {
      q.x0 = x0[i]; q.y0 = y0[i]; q.dx = dx[i]; q.dy = dy[i];
      q.w = (im.w - q.x0 + q.dx - 1) / q.dx; q.h = (im.h - q.y0 + q.dy - 1) / q.dy;
    } else { q.x0 = q.y0 = 0; q.dx = q.dy = 1; q.w = im.w; q.h = im.h; }
    if (q.w <= 0 || q.h <= 0) continue;
    q.stride = ((size_t)q.w * bits + 7) / 8;
    q.off = total;
    total += (q.stride + 1) * q.h;
    passes.push_back(q);
  }
  std::vector<uint8_t> raw(total);
  uLongf rawlen = raw.size();
  if (uncompress(raw.data(), &rawlen, z.data(), z.size()) != Z_OK || rawlen != raw.size()) {
    err = "PNG inflate failed"; return false;
  }
  const bool key = ctype == 2 && have_trns;
  im.comp = ctype == 3 ? (have_trns ? 4 : 3) : (key ? 4 : chan);
  im.px.assign((size_t)im.w * im.h * im.comp, 0);
  const int bpp = std::max(1, bits / 8);
  const int grey_scale = depth == 1 ? 0xff : depth == 2 ? 0x55 : depth == 4 ? 0x11 : 1;
  for (int i = 0; i < n_trns && ctype == 3; ++i) pal[4 * i + 3] = trns[i];
  for (const Pass& q : passes) {
    uint8_t* sub = raw.data() + q.off;
    if (!png_unfilter(sub, q.stride, q.h, bpp)) { err = "bad PNG filter"; return false; }
    for (int y = 0; y < q.h; ++y) {
      const uint8_t* row = sub + (q.stride + 1) * y + 1;
      for (int x = 0; x < q.w; ++x) {
        uint8_t* out = im.px.data() + ((size_t)(q.y0 + y * q.dy) * im.w + (q.x0 + x * q.dx)) * im.comp;
        if (depth < 8) {
          const int per = 8 / depth;
          const int v = (row[x / per] >> (8 - depth * (x % per + 1))) & ((1 << depth) - 1);
          if (ctype == 3) memcpy(out, pal + 4 * v, im.comp);
          else out[0] = (uint8_t)(v * grey_scale);
        } else if (ctype == 3) {
          memcpy(out, pal + 4 * row[x], im.comp);
        } else {
          const int step = depth / 8;
          const uint8_t* s = row + (size_t)x * chan * step;
          for (int c = 0; c < chan; ++c) out[c] = s[c * step];  // 16 bit: the high byte
          if (key) {  // tRNS holds the key as three big-endian 16-bit samples
            const bool hit = step == 2 ? memcmp(s, trns, 6) == 0
                                       : (s[0] == trns[1] && s[1] == trns[3] && s[2] == trns[5]);
            out[3] = hit ? 0 : 255;
          }
        }
      }
    }
  }
  return true;
}

}  // namespace

bool load_mesh_file(const std::string& path, RawMesh& out, std::string& err) {
  if (ends_with(path, ".ply")) return load_ply(path, out, err);
  return load_obj(path, out, err);
}

// Texture load as the reference does it (imageio.cpp:12-60, texture.h:11-23):
// rows flipped so row 0 is the bottom of the image, sRGB -> linear with
// pow(x, 2.2), then quantised back to 8 bits by truncation.
bool load_texture_file(const std::string& path, Texture& out, std::string& err) {
  std::string data;
  if (!read_file(path, data)) { err = "cannot open texture \"" + path + "\""; return false; }
  Image8 im;
  if (!load_ppm(data, im)) {
    std::string perr;
    // by content, in stb_image's order: JPEG, PNG, BMP, and TGA (which has no signature) last
    const bool jpeg = data.size() > 2 && (uint8_t)data[0] == 0xff && (uint8_t)data[1] == 0xd8;
    const bool png = data.size() > 4 && (uint8_t)data[0] == 0x89 && data[1] == 'P';
    const bool bmp = data.size() > 2 && data[0] == 'B' && data[1] == 'M';
    const bool ok = jpeg ? load_jpeg(data, im, perr) : png ? load_png(data, im, perr) : bmp ? load_bmp(data, im, perr) : load_tga(data, im, perr);
    if (!ok) { err = "texture \"" + path + "\": " + perr; return false; }
  }
  out.w = im.w; out.h = im.h;
  out.rgba.resize((size_t)im.w * im.h * 4);
  const float inv = 1.f / 255.f;
  for (int y = 0; y < im.h; ++y) {
    for (int x = 0; x < im.w; ++x) {
      const uint8_t* src = im.px.data() + ((size_t)(im.h - 1 - y) * im.w + x) * im.comp;
      float c[4];
      if (im.comp == 1) { c[0] = c[1] = c[2] = src[0] * inv; c[3] = 1.f; }
      else if (im.comp == 3) { c[0] = src[0] * inv; c[1] = src[1] * inv; c[2] = src[2] * inv; c[3] = 1.f; }
      else { c[0] = src[0] * inv; c[1] = src[1] * inv; c[2] = src[2] * inv; c[3] = src[3] * inv; }
      uint8_t* dst = out.rgba.data() + ((size_t)y * im.w + x) * 4;
      dst[0] = (uint8_t)(powf(c[0], 2.2f) * 255);
      dst[1] = (uint8_t)(powf(c[1], 2.2f) * 255);
      dst[2] = (uint8_t)(powf(c[2], 2.2f) * 255);
      dst[3] = (uint8_t)(c[3] * 255);
    }
  }
  return true;
}

namespace {

float half_to_float(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 1023u, bits;
  if (e == 0) {
    if (m == 0) bits = sign;
    else {  // subnormal half: normalise
      int shift = 0;
      while (!(m & 1024u)) { m <<= 1; ++shift; }
      bits = sign | ((uint32_t)(113 - shift) << 23) | ((m & 1023u) << 13);
    }
  } else if (e == 31) bits = sign | 0x7f800000u | (m << 13);
  else bits = sign | ((e + 112u) << 23) | (m << 13);
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

// PFM (PF, 3 x float32, bottom row first when the scale is negative = little endian).
bool load_pfm(const std::string& data, Infinite& out, std::string& err) {
  int w = 0, h = 0; float scale = 0.f; int consumed = 0;
  if (sscanf(data.c_str() + 2, "%d %d %f%n", &w, &h, &scale, &consumed) != 3 || w <= 0 || h <= 0) {
    err = "bad PFM header"; return false;
  }
  size_t p = 2 + (size_t)consumed + 1;
  if (scale > 0.f) { err = "big-endian PFM is not supported"; return false; }
  if (p + (size_t)w * h * 12 > data.size()) { err = "truncated PFM"; return false; }
  out.w = w; out.h = h;
  out.data.resize((size_t)w * h);
  for (int y = 0; y < h; ++y)
    memcpy(&out.data[(size_t)(h - 1 - y) * w], data.data() + p + (size_t)y * w * 12, (size_t)w * 12);
  return true;
}

// Single-part scanline OpenEXR with channels R, G, B (half, float or uint; other channels are
// skipped) stored uncompressed or with RLE / ZIPS / ZIP compression: what tinyexr's LoadEXR hands to
// the reference (imageio.cpp:81-103) for such a file, rows in the file's y order.  Tiled and
// multi-part files and the PIZ / PXR24 / B44 / DWA codecs are refused with a message.
struct ExrIn {
  const uint8_t* p; size_t n, at = 0; bool ok = true;
  bool need(size_t k) { if (at + k > n) ok = false; return ok; }
  std::string str() { std::string s; while (need(1) && p[at]) s.push_back((char)p[at++]); if (ok) ++at; return s; }
  int32_t i32() { int32_t v = 0; if (need(4)) { memcpy(&v, p + at, 4); at += 4; } return v; }
  uint64_t u64() { uint64_t v = 0; if (need(8)) { memcpy(&v, p + at, 8); at += 8; } return v; }
};
bool load_exr(const std::string& data, Infinite& out, std::string& err) {
  ExrIn in{(const uint8_t*)data.data(), data.size()};
  in.i32();
  int32_t version = in.i32();
  if ((version & 0xff) != 2 || (version & 0x200) || (version & 0x1000) || (version & 0x800)) {
    err = "only single-part scanline OpenEXR 2 files are read"; return false;
  }
  struct Chan { std::string name; int type; };
  std::vector<Chan> chans;
  int compression = -1, x0 = 0, y0 = 0, x1 = -1, y1 = -1;
  for (;;) {
    std::string name = in.str();
    if (!in.ok) { err = "truncated EXR header"; return false; }
    if (name.empty()) break;
    std::string type = in.str();
    int32_t size = in.i32();
    if (!in.ok || size < 0 || !in.need((size_t)size)) { err = "truncated EXR header"; return false; }
    size_t end = in.at + (size_t)size;
    if (name == "channels") {
      while (in.at < end && in.p[in.at]) {
        Chan c;
        c.name = in.str();
        c.type = in.i32();
        in.at += 4;  // pLinear, reserved
        int32_t xs = in.i32(), ys = in.i32();
        if (xs != 1 || ys != 1) { err = "subsampled EXR channels are not supported"; return false; }
        if (c.type < 0 || c.type > 2) { err = "bad EXR channel type"; return false; }
        chans.push_back(c);
      }
    } else if (name == "compression") {
      compression = in.p[in.at];
    } else if (name == "dataWindow") {
      x0 = in.i32(); y0 = in.i32(); x1 = in.i32(); y1 = in.i32();
    }
    in.at = end;
  }
  const long long w64 = (long long)x1 - x0 + 1, h64 = (long long)y1 - y0 + 1;
  if (w64 <= 0 || h64 <= 0 || w64 * h64 > (1ll << 28) || chans.empty()) { err = "EXR header lacks channels or a data window"; return false; }
  const int w = (int)w64, h = (int)h64;
  if (compression < 0 || compression > 3) {
    static const char* names[] = {"NONE", "RLE", "ZIPS", "ZIP", "PIZ", "PXR24", "B44", "B44A", "DWAA", "DWAB"};
    err = std::string("EXR compression ") + (compression >= 0 && compression < 10 ? names[compression] : "?") +
          " is not supported (NONE, RLE, ZIPS and ZIP are)";
    return false;
  }
  int chan_of[3] = {-1, -1, -1};
  size_t line_bytes = 0;
  std::vector<size_t> chan_off(chans.size());
  for (size_t c = 0; c < chans.size(); ++c) {
    chan_off[c] = line_bytes;
    line_bytes += (size_t)w * (chans[c].type == 1 ? 2 : 4);
    if (chans[c].name == "R") chan_of[0] = (int)c;
    if (chans[c].name == "G") chan_of[1] = (int)c;
    if (chans[c].name == "B") chan_of[2] = (int)c;
  }
  if (chan_of[0] < 0 && chans.size() == 1) chan_of[0] = chan_of[1] = chan_of[2] = 0;  // luminance only
  if (chan_of[0] < 0 || chan_of[1] < 0 || chan_of[2] < 0) { err = "EXR file has no R, G, B channels"; return false; }
  const int lines_per_chunk = compression == 3 ? 16 : 1;
  const int n_chunks = (h + lines_per_chunk - 1) / lines_per_chunk;
  std::vector<uint64_t> offsets((size_t)n_chunks);
  for (auto& o : offsets) o = in.u64();
  if (!in.ok) { err = "truncated EXR offset table"; return false; }
  out.w = w; out.h = h;
  out.data.assign((size_t)w * h, f3{0.f, 0.f, 0.f});
  std::vector<uint8_t> raw, tmp;
  for (int k = 0; k < n_chunks; ++k) {
    ExrIn ch{in.p, in.n, (size_t)offsets[k]};
    int32_t y = ch.i32(), size = ch.i32();
    if (!ch.ok || size < 0 || !ch.need((size_t)size) || y < y0 || y > y1) { err = "bad EXR chunk"; return false; }
    const int lines = std::min(lines_per_chunk, y1 - y + 1);
    const size_t want = line_bytes * (size_t)lines;
    const uint8_t* src = ch.p + ch.at;
    if (compression != 0 && (size_t)size < want) {
      tmp.assign(want, 0);
      if (compression == 1) {  // run-length: n >= 0 repeats the next byte n + 1 times, n < 0 copies -n bytes
        size_t i = 0, o = 0;
        while (i < (size_t)size && o < want) {
          int n = (int8_t)src[i++];
          if (n < 0) { size_t c = (size_t)(-n); if (i + c > (size_t)size || o + c > want) break; memcpy(&tmp[o], src + i, c); i += c; o += c; }
          else { size_t c = (size_t)n + 1; if (i >= (size_t)size || o + c > want) break; memset(&tmp[o], src[i++], c); o += c; }
        }
        if (o != want) { err = "bad run-length data in EXR chunk"; return false; }
      } else {
        uLongf got = (uLongf)want;
        if (uncompress(tmp.data(), &got, src, (uLong)size) != Z_OK || got != want) { err = "bad zlib data in EXR chunk"; return false; }
      }
      for (size_t i = 1; i < want; ++i) tmp[i] = (uint8_t)(tmp[i - 1] + tmp[i] - 128);  // undo the byte predictor
      raw.resize(want);
      const size_t half = (want + 1) / 2;                                                 // and the even/odd split
      for (size_t i = 0; i < want; ++i) raw[i] = (i & 1) ? tmp[half + i / 2] : tmp[i / 2];
      src = raw.data();
    } else if ((size_t)size != want) {
      err = "bad EXR chunk size"; return false;
    }
    for (int l = 0; l < lines; ++l) {
      const uint8_t* line = src + (size_t)l * line_bytes;
      f3* dst = &out.data[(size_t)(y - y0 + l) * w];
      for (int c = 0; c < 3; ++c) {
        const Chan& cc = chans[(size_t)chan_of[c]];
        const uint8_t* q = line + chan_off[(size_t)chan_of[c]];
        for (int x = 0; x < w; ++x) {
          float v;
          if (cc.type == 1) { uint16_t hv; memcpy(&hv, q + 2 * x, 2); v = half_to_float(hv); }
          else if (cc.type == 2) memcpy(&v, q + 4 * x, 4);
          else { uint32_t u; memcpy(&u, q + 4 * x, 4); v = (float)u; }
          (c == 0 ? dst[x].x : c == 1 ? dst[x].y : dst[x].z) = v;
        }
      }
    }
  }
  return true;
}

}  // namespace

// Environment map: OpenEXR as the reference reads it (parsescene.cpp:551-588, imageio.cpp:81-103),
// or PFM.  Stored top row first, the row order tinyexr hands to the reference.
bool load_envmap_file(const std::string& path, Infinite& out, std::string& err) {
  std::string data;
  if (!read_file(path, data)) { err = "cannot open environment map \"" + path + "\""; return false; }
  std::string why;
  bool ok = false;
  if (data.size() >= 8 && (uint8_t)data[0] == 0x76 && (uint8_t)data[1] == 0x2f && (uint8_t)data[2] == 0x31 && (uint8_t)data[3] == 0x01)
    ok = load_exr(data, out, why);
  else if (data.size() >= 3 && data[0] == 'P' && data[1] == 'F')
    ok = load_pfm(data, out, why);
  else
    why = "must be an OpenEXR or a PFM (PF) file";
  if (!ok) { err = "environment map \"" + path + "\": " + why; return false; }
  out.valid = 1;
  return true;
}

}  // namespace caster
