"""Time per Render()-style call: one iteration per call, film read back every call (the reference's
interactive loop, CudaRender.cpp:44-60), against one iteration per call without the read-back."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_scenes
caster = importlib.import_module("cuda-pathtracer_amd")
for name, w, h, detail in (("cornell_pt", 512, 512, 1.0), ("car_pt", 1920, 1080, 1.0)):
    path = gen_scenes.generate("/tmp/lat_" + name, name, width=w, height=h, detail=detail)
    sc = caster.Scene(path)
    r = caster.Renderer(sc, 0, 1, 1)
    r.render(1, 4, True)
    for read in (False, True):
        n = 60
        t0 = time.perf_counter()
        for it in range(n):
            r.render(5 + it, 1, False)
            if read:
                r.read_film(1)
        dt = (time.perf_counter() - t0) / n
        print("%-12s %dx%d  %.3f ms per iteration%s  (%.1f Msamples/s)" % (name, w, h, dt * 1e3, " + read-back" if read else "", w * h / dt / 1e6), flush=True)
    r.close()
