"""Experiment: streams x iterations-in-flight inside one renderer.  Arguments: SxI ..."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench
caster = importlib.import_module("cuda-pathtracer_amd")
path = bench.scene_path("car_pt", 1.0, 1920, 1080, 8, 0, lambda: None)
scene = caster.Scene(path)
def run(n_s, iif, steps=4):
    r = caster.Renderer(scene, 0, iif, n_s)
    spp = n_s * iif
    r.render(1, spp, False)
    t0 = time.perf_counter()
    for k in range(steps):
        r.render_async(1 + (k + 1) * spp, spp, False)
    r.sync()
    dt = time.perf_counter() - t0
    n = steps * spp * 1920 * 1080
    print("streams %d x %d iterations: %.1f Msamples/s" % (n_s, iif, n / dt / 1e6), flush=True)
    r.close()
for a in sys.argv[1:]:
    n, s = a.split("x"); run(int(n), int(s))
