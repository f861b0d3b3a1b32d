"""Rays and kernel time per bounce: the benchmark scene rendered with maxDepth 1, 2, ... on one stream with the
per-class kernel timers on; the differences between consecutive depths are the launches of one bounce (paths are
the same up to the depth they are cut at).  Prints, per bounce, extension rays and their rate and shadow + probe
rays and theirs."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_scenes
caster = importlib.import_module("cuda-pathtracer_amd")
prev = None
for depth in range(1, 7):
    path = gen_scenes.generate("/tmp/pb_%d" % depth, "car_pt", 1.0, 1920, 1080, depth)
    sc = caster.Scene(path)
    r = caster.Renderer(sc, 0, 64, 1)
    r.render(1, 64, True)
    r.reset_stats()
    r.set_profiling(True)
    r.render(65, 64, False)
    st, kt = r.stats(), r.kernel_times()
    r.close()
    cur = (st["rays"][0], st["rays"][1] + st["rays"][2], kt["extend"][0], kt["shadow"][0] + kt["probe"][0], kt["shade"][0])
    d = cur if prev is None else tuple(a - b for a, b in zip(cur, prev))
    print("bounce %d: extension %6.1f M rays %6.2f ms %5.2f Mrays/ms | shadow+probe %6.1f M rays %6.2f ms %5.2f Mrays/ms | shade %6.2f ms" % (
        depth - 1, d[0] / 1e6, d[2], d[0] / 1e6 / max(d[2], 1e-9), d[1] / 1e6, d[3], d[1] / 1e6 / max(d[3], 1e-9), d[4]), flush=True)
    prev = cur
