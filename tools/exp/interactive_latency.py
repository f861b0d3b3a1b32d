"""Time per Render()-style call: one iteration per call, film read back every call (the reference's
interactive loop, CudaRender.cpp:44-60), against one iteration per call without the read-back, and both
with a look-ahead window (caster_set_lookahead): mean, median and worst call."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_scenes
caster = importlib.import_module("cuda-pathtracer_amd")
for name, w, h, detail in (("cornell_pt", 512, 512, 1.0), ("car_pt", 1920, 1080, 1.0)):
    path = gen_scenes.generate("/tmp/lat_" + name, name, width=w, height=h, detail=detail)
    sc = caster.Scene(path)
    for look in (0, 8, 16, 64):
        r = caster.Renderer(sc, 0, max(look, 1), 1)
        if look:
            r.set_lookahead(look)
        r.render(1, 4, True)
        for read in (False, True):
            n = 192
            times = []
            for it in range(n):
                t0 = time.perf_counter()
                r.render(5 + it + (n if read else 0), 1, False)
                if read:
                    r.read_film(1)
                times.append(time.perf_counter() - t0)
            t = np.array(times) * 1e3
            print("%-12s %dx%d look-ahead %2d  mean %.3f ms per iteration%s, median %.3f, worst %.2f  (%.1f Msamples/s)" % (
                name, w, h, look, t.mean(), " + read-back" if read else "", np.median(t), t.max(), w * h / t.mean() / 1e3), flush=True)
        r.close()
