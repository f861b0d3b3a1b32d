"""Experiment: shade time per vertex with the scene's mixed materials against one material type
everywhere (how much of the shade stage is divergence between material types)."""
import importlib, json, os, shutil, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench
caster = importlib.import_module("cuda-pathtracer_amd")
path = bench.scene_path(sys.argv[1] if len(sys.argv) > 1 else "car_pt", 1.0, 1920, 1080, 8, 0, lambda: None)
base = os.path.dirname(path)
doc = json.load(open(path))
def run(tag, d):
    p = os.path.join(base, "exp_%s.json" % tag)
    json.dump(d, open(p, "w"))
    sc = caster.Scene(p, build_device=0)
    r = caster.Renderer(sc, 0, 16, 1)
    r.render(1, 16, True); r.reset_stats(); r.set_profiling(True)
    t = time.perf_counter(); r.render(17, 32, False); dt = time.perf_counter() - t
    st, kt = r.stats(), r.kernel_times()
    v = st["rays"][0]
    print("%-14s %.0f Msamples/s, %.2f vertices/sample, shade %.3f ns/vertex, trace %.3f ns/ray" % (
        tag, st["samples"] / dt / 1e6, v / st["samples"], kt["shade"][0] * 1e6 / v,
        (kt["extend"][0] + kt["shadow"][0] + kt["probe"][0]) * 1e6 / sum(st["rays"])), flush=True)
    r.close(); sc.close()
run("mixed", doc)
for kind, extra in (("lambertian", {}), ("roughconduct", {"alpha": 0.1, "eta": [0.2, 0.92, 1.1], "k": [3.9, 2.45, 2.14]}),
                    ("substrate", {"alpha": 0.05})):
    d = json.loads(json.dumps(doc))
    for m in d["material"]:
        keep = {k: m[k] for k in ("name", "diffuse", "specular") if k in m}
        m.clear(); m.update(keep); m["bsdf"] = kind; m.update(extra)
    run(kind, d)
