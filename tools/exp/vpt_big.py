"""Experiment: the volumetric integrator on the 1.37 M-triangle scene (no media): how much the
in-kernel walks of k_shade_vpt cost when the tree is deep."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench
caster = importlib.import_module("cuda-pathtracer_amd")
path = bench.scene_path("car_pt", 1.0, 1920, 1080, 8, 0, lambda: None)
doc = json.load(open(path)); doc["integrator"] = "vpt"
p = os.path.join(os.path.dirname(path), "exp_vpt.json"); json.dump(doc, open(p, "w"))
sc = caster.Scene(p, build_device=0)
r = caster.Renderer(sc, 0, 8, 2)
r.render(1, 16, True); r.reset_stats(); r.set_profiling(True)
t = time.perf_counter(); r.render(17, 32, False); dt = time.perf_counter() - t
st, kt = r.stats(), r.kernel_times()
print("%s: %.1f Msamples/s" % (os.environ.get("CASTER_LIB", "default")[-20:], st["samples"] / dt / 1e6), {k: round(v[0]) for k, v in kt.items()}, flush=True)
