"""Experiment: do two renderers (two HIP streams) in ONE process overlap well enough to
beat one renderer with the same total number of paths in flight?"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench
caster = importlib.import_module("cuda-pathtracer_amd")
path = bench.scene_path("car_pt", 1.0, 1920, 1080, 8, 0, lambda: None)
scene = caster.Scene(path)
def run(n_r, spp_each, steps=6):
    rs = [caster.Renderer(scene, 0, spp_each) for _ in range(n_r)]
    it = 1
    for r in rs:
        r.render_async(it, spp_each, False); it += spp_each
    for r in rs: r.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        for r in rs:
            r.render_async(it, spp_each, False); it += spp_each
    for r in rs: r.sync()
    dt = time.perf_counter() - t0
    n = steps * n_r * spp_each * 1920 * 1080
    print("renderers %d x %d spp: %.1f Msamples/s" % (n_r, spp_each, n / dt / 1e6), flush=True)
    for r in rs: r.close()
import sys as _s
for a in _s.argv[1:]:
    n, s = a.split("x"); run(int(n), int(s))
