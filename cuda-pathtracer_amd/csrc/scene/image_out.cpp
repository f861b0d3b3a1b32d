// Image writers for the film: 8-bit PNG as the reference's ImageIO::SavePng
// (CUDA-PathTracer/tracer/imageio.cpp:62-79: rows flipped so that the top
// scanline comes first, channel = unsigned(clamp(v, 0, 1) * 255)), OpenEXR with
// half-float B, G, R channels as ImageIO::SaveExr asks of tinyexr
// (imageio.cpp:105-163; written here as an uncompressed scanline file), and
// PFM (raw float32, bottom row first, which is the film's own row order).
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "scene.h"

namespace caster {

static void put_u32be(std::vector<uint8_t>& v, uint32_t x) {
  v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}
static void png_chunk(std::vector<uint8_t>& out, const char* tag, const std::vector<uint8_t>& data) {
  put_u32be(out, (uint32_t)data.size());
  size_t at = out.size();
  out.insert(out.end(), tag, tag + 4);
  out.insert(out.end(), data.begin(), data.end());
  uint32_t crc = (uint32_t)crc32(0L, out.data() + at, (uInt)(out.size() - at));
  put_u32be(out, crc);
}
static bool write_file(const std::string& path, const void* data, size_t n, std::string& err) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) { err = "cannot open " + path + " for writing"; return false; }
  bool ok = fwrite(data, 1, n, f) == n;
  ok = (fclose(f) == 0) && ok;
  if (!ok) err = "short write to " + path;
  return ok;
}

// rgb: W*H*3 floats, row 0 = bottom scanline.
bool write_png(const std::string& path, int w, int h, const float* rgb, std::string& err) {
  std::vector<uint8_t> raw((size_t)h * (1 + (size_t)w * 3));
  for (int i = 0; i < h; ++i) {
    uint8_t* row = &raw[(size_t)i * (1 + (size_t)w * 3)];
    row[0] = 0;  // filter: none
    const float* src = rgb + (size_t)(h - i - 1) * w * 3;
    for (int k = 0; k < w * 3; ++k) {
      float v = src[k];
      v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);  // NaN falls through to the conversion below as 0
      if (!(v == v)) v = 0.f;
      row[1 + k] = (uint8_t)(unsigned)(v * 255.f);
    }
  }
  uLongf cap = compressBound((uLong)raw.size());
  std::vector<uint8_t> z(cap);
  if (compress2(z.data(), &cap, raw.data(), (uLong)raw.size(), 6) != Z_OK) { err = "zlib compress failed"; return false; }
  z.resize(cap);
  std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  std::vector<uint8_t> ihdr;
  put_u32be(ihdr, (uint32_t)w); put_u32be(ihdr, (uint32_t)h);
  ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);  // 8-bit RGB
  png_chunk(out, "IHDR", ihdr);
  png_chunk(out, "IDAT", z);
  png_chunk(out, "IEND", {});
  return write_file(path, out.data(), out.size(), err);
}

bool write_pfm(const std::string& path, int w, int h, const float* rgb, std::string& err) {
  char head[64];
  int n = snprintf(head, sizeof head, "PF\n%d %d\n-1.0\n", w, h);  // negative scale: little endian
  std::vector<uint8_t> out(head, head + n);
  const uint8_t* p = (const uint8_t*)rgb;
  out.insert(out.end(), p, p + (size_t)w * h * 3 * sizeof(float));
  return write_file(path, out.data(), out.size(), err);
}

// IEEE binary32 -> binary16, round to nearest even, overflow to infinity.
uint16_t float_to_half(float f) {
  uint32_t x; memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t mag = x & 0x7fffffffu;
  if (mag >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((mag > 0x7f800000u) ? 0x200u : 0u));  // inf / NaN
  if (mag >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);  // rounds to >= 65520: infinity
  if (mag < 0x33000001u) return (uint16_t)sign;               // rounds to zero (<= 2^-25)
  int exp = (int)(mag >> 23) - 127;
  uint32_t man = (mag & 0x7fffffu) | 0x800000u;
  int shift = exp < -14 ? 13 + (-14 - exp) : 13;  // subnormal halves lose more bits
  uint32_t half_man = man >> shift;
  uint32_t rem = man & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
  if (rem > halfway || (rem == halfway && (half_man & 1u))) ++half_man;
  uint32_t bits = exp < -14 ? half_man : (((uint32_t)(exp + 15) << 10) + (half_man - 0x400u));
  return (uint16_t)(sign | bits);
}

namespace {
struct ExrOut {
  std::vector<uint8_t> b;
  void raw(const void* p, size_t n) { const uint8_t* q = (const uint8_t*)p; b.insert(b.end(), q, q + n); }
  void str(const char* s) { raw(s, strlen(s) + 1); }
  void i32(int32_t v) { raw(&v, 4); }
  void f32(float v) { raw(&v, 4); }
  void attr(const char* name, const char* type, int32_t size) { str(name); str(type); i32(size); }
};
}  // namespace

// Uncompressed scanline OpenEXR, channels B, G, R as half, top scanline first.
bool write_exr(const std::string& path, int w, int h, const float* rgb, std::string& err) {
  ExrOut o;
  o.i32(20000630); o.i32(2);
  const char* names[3] = {"B", "G", "R"};
  o.attr("channels", "chlist", 3 * (2 + 16) + 1);
  for (int c = 0; c < 3; ++c) { o.str(names[c]); o.i32(1 /*HALF*/); o.i32(0); o.i32(1); o.i32(1); }
  o.b.push_back(0);
  o.attr("compression", "compression", 1); o.b.push_back(0);
  o.attr("dataWindow", "box2i", 16); o.i32(0); o.i32(0); o.i32(w - 1); o.i32(h - 1);
  o.attr("displayWindow", "box2i", 16); o.i32(0); o.i32(0); o.i32(w - 1); o.i32(h - 1);
  o.attr("lineOrder", "lineOrder", 1); o.b.push_back(0);
  o.attr("pixelAspectRatio", "float", 4); o.f32(1.f);
  o.attr("screenWindowCenter", "v2f", 8); o.f32(0.f); o.f32(0.f);
  o.attr("screenWindowWidth", "float", 4); o.f32(1.f);
  o.b.push_back(0);
  const uint64_t line_bytes = 8 + (uint64_t)w * 3 * 2;
  uint64_t first = o.b.size() + (uint64_t)h * 8;
  for (int y = 0; y < h; ++y) { uint64_t off = first + (uint64_t)y * line_bytes; o.raw(&off, 8); }
  std::vector<uint16_t> line((size_t)w * 3);
  const int src_channel[3] = {2, 1, 0};  // B, G, R from r, g, b
  for (int y = 0; y < h; ++y) {
    const float* src = rgb + (size_t)(h - y - 1) * w * 3;
    for (int c = 0; c < 3; ++c)
      for (int x = 0; x < w; ++x) line[(size_t)c * w + x] = float_to_half(src[(size_t)x * 3 + src_channel[c]]);
    o.i32(y); o.i32((int32_t)(w * 3 * 2));
    o.raw(line.data(), line.size() * 2);
  }
  return write_file(path, o.b.data(), o.b.size(), err);
}

bool write_image(const std::string& path, int w, int h, const float* rgb, std::string& err) {
  if (w <= 0 || h <= 0 || !rgb) { err = "write_image: empty image"; return false; }
  auto ends_with = [&](const char* ext) {
    size_t n = strlen(ext);
    if (path.size() < n) return false;
    for (size_t i = 0; i < n; ++i)
      if (tolower((unsigned char)path[path.size() - n + i]) != ext[i]) return false;
    return true;
  };
  if (ends_with(".png")) return write_png(path, w, h, rgb, err);
  if (ends_with(".exr")) return write_exr(path, w, h, rgb, err);
  if (ends_with(".pfm")) return write_pfm(path, w, h, rgb, err);
  err = "write_image: unknown extension (png, exr, pfm): " + path;
  return false;
}

}  // namespace caster
