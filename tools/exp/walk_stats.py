"""Experiment (needs a library built with -DCASTER_WALK_STATS, given through CASTER_LIB):
lane utilisation breakdown of the traversal kernel."""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench
caster = importlib.import_module("cuda-pathtracer_amd")
path = bench.scene_path("car_pt", 1.0, 1920, 1080, 8, 0, lambda: None)
scene = caster.Scene(path)
r = caster.Renderer(scene, 0, 8, 1)
buf = (C.c_ulonglong * 32)()
caster.lib().caster_debug_walk_stats(buf)
r.render(1, 8, True)
caster.lib().caster_debug_walk_stats(buf)
st = r.stats()
for a, name in ((0, "closest"), (1, "any")):
    v = list(buf[a * 16:a * 16 + 16])
    steps = v[0] + v[2]
    print(name, "node steps %d (%.1f lanes), prim steps %d (%.1f lanes), busy lanes/step %.1f, lanes holding both %.1f, "
          "outer iterations %d" % (v[0], v[1] / max(v[0], 1), v[2], v[3] / max(v[2], 1), v[4] / max(steps, 1),
                                   v[5] / max(steps, 1), v[6]))
    print("   node steps with a lane on the careful push path %.1f %% (%.1f lanes of them); pops: %.2f entries per call, "
          "%d calls; spilled pushes %d, pops from the spill %d, careful pushes (lanes) %d" % (
              100.0 * v[5] / max(v[0], 1), v[7] / max(v[5], 1), v[9] / max(v[8], 1), v[8], v[10], v[11], v[12]))
    print("   primitive steps: lanes whose leaf goes on %.1f, lanes without a leaf %.1f, pairs of the two %.1f per step" % (
        v[13] / max(v[2], 1), v[15] / max(v[2], 1), v[14] / max(v[2], 1)))
print("rays", st["rays"])
