// Mutation fuzzer for the file readers of the scene loader (scene files, textures, environment maps,
// meshes).  A scene-file seed must lie in FUZZ_TMP beside the files it names: its mutations are written there.
// Built with the host sanitizers (see run.sh); every input either loads or is refused with a
// message, and the sanitizers abort on any out-of-bounds access on the way.
//   fuzz_loaders <seed file>... : mutates each seed `rounds` times (FUZZ_ROUNDS, default 2000)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <vector>

#include "../../cuda-pathtracer_amd/csrc/scene/scene.h"

static std::string slurp(const char* path) {
  std::ifstream f(path, std::ios::binary);
  return std::string((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
  const int rounds = getenv("FUZZ_ROUNDS") ? atoi(getenv("FUZZ_ROUNDS")) : 2000;
  const char* tmp = getenv("FUZZ_TMP") ? getenv("FUZZ_TMP") : "/tmp";
  std::mt19937 rng(12345);
  long loaded = 0, refused = 0;
  for (int a = 1; a < argc; ++a) {
    const std::string seed = slurp(argv[a]);
    const char* dot = strrchr(argv[a], '.');
    const std::string ext = dot ? dot : "";
    const std::string path = std::string(tmp) + "/fuzz_case" + ext;
    for (int r = 0; r < rounds; ++r) {
      std::string d = seed;
      const int kind = (int)(rng() % 4);
      const int edits = 1 + (int)(rng() % 8);
      for (int e = 0; e < edits && !d.empty(); ++e) {
        size_t at = rng() % d.size();
        if (kind == 0) d[at] = (char)(rng() & 255);                       // byte noise
        else if (kind == 1) d[at] ^= (char)(1u << (rng() % 8));           // bit flips
        else if (kind == 2 && e == 0) d.resize(at);                       // truncation
        else if (kind == 3) d[at] = (char)((rng() & 1) ? 0xff : 0x00);    // extremes
      }
      { std::ofstream f(path, std::ios::binary); f.write(d.data(), (std::streamsize)d.size()); }
      std::string err;
      bool ok;
      try {
        if (ext == ".exr" || ext == ".pfm") { caster::Infinite inf; ok = caster::load_envmap_file(path, inf, err); }
        else if (ext == ".obj" || ext == ".ply") { caster::RawMesh m; ok = caster::load_mesh_file(path, m, err); }
        else if (ext == ".json") { caster::Scene sc; ok = caster::load_scene(path.c_str(), sc, err); }
        else { caster::Texture t; ok = caster::load_texture_file(path, t, err); }
      } catch (const std::exception& ex) {   // allocation failures and the like: a refusal as well
        ok = false;
      }
      ok ? ++loaded : ++refused;
    }
    printf("%-28s %d mutations: %ld loaded, %ld refused so far\n", argv[a], rounds, loaded, refused);
    fflush(stdout);
  }
  return 0;
}
