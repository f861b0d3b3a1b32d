"""Load time of the benchmark scene with the BVH built on the host and on the GPU."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench
os.environ["CASTER_TIMING"] = "1"
caster = importlib.import_module("cuda-pathtracer_amd")
path = bench.scene_path("car_pt", 1.0, 1920, 1080, 8, 0, lambda: None)
for dev in (None, 0, 0, None):
    t = time.perf_counter()
    s = caster.Scene(path, build_device=dev)
    print("build on %s: %.3f s total" % ("host" if dev is None else "gpu", time.perf_counter() - t), flush=True)
    s.close()
