#!/usr/bin/env python3
"""Regenerates the golden fixtures in this directory from the CPU oracle.

The reference ships no tests, scenes or golden vectors, so these fixtures are
produced by the oracle itself (parity UNPINNED by the reference): they pin the
oracle against regressions and give the HIP renderer fixed targets.  The only
externally published value is Thrust's minstd_rand known answer (the 10000th
output of a default-constructed engine is 399268537).

  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_scenes  # noqa: E402
import oracle_lib as ol  # noqa: E402

SCENES = {
    "cornell_ao": dict(name="cornell_ao", width=64, height=64),
    "cornell_pt": dict(name="cornell_pt", width=64, height=64),
    "cornell_vpt": dict(name="cornell_vpt", width=64, height=64, depth=8),
    "stairs_pt": dict(name="stairs_pt", width=80, height=48, detail=0.04),
    "car_env": dict(name="car_pt", width=80, height=48, detail=0.004, env=True),
}
SPP = 8
# Light tracing and bidirectional path tracing: a film and a first-hit map only (the per-function vectors above are those of the
# camera-side integrators).  The film's rounding depends on the oracle's thread count: fixed at 4.
LT_SCENES = {
    "cornell_fog_lt": dict(name="cornell_fog_lt", width=64, height=64, depth=8),
    "cornell_fog_bdpt": dict(name="cornell_fog_bdpt", width=48, height=48),
}
LT_SPP, LT_THREADS = 4, 4


def f32list(a):
    return [float(x) for x in np.asarray(a, np.float32).ravel()]


def main():
    vec = {"rng": {}, "scenes": {}}
    vec["rng"]["minstd_10000th_seed1"] = int(ol.rng_raw(1, 10000)[-1])
    vec["rng"]["wang_hash"] = {str(v): int(ol.wang_hash(v)) for v in (0, 1, 2, 61, 12345, 0xffffffff)}
    seed = (ol.wang_hash(0) + ol.wang_hash(1)) & 0xffffffff
    vec["rng"]["seed_pixel0_iter1"] = int(seed)
    vec["rng"]["first8_floats_pixel0_iter1"] = f32list(ol.rng_floats(seed, 8))
    vec["rng"]["first8_raw_pixel0_iter1"] = [int(x) for x in ol.rng_raw(seed, 8)]
    for key, kw in SCENES.items():
        kw = dict(kw)
        name = kw.pop("name")
        d = os.path.join(HERE, "scenes", key)
        path = gen_scenes.generate(d, name, **kw)
        o = ol.Oracle(path)
        c = ol.Counts()
        acc = o.render(1, SPP, 8, counts=c)
        np.save(os.path.join(HERE, key + "_sum.npy"), acc)
        np.save(os.path.join(HERE, key + "_ids.npy"), o.primary_ids(1))
        sv = {"spp": SPP, "counts": c.as_dict(), "n_prims": o.n_prims, "n_nodes": o.n_nodes,
              "n_lights": o.n_lights}
        W, H = o.width, o.height
        px = [(0, 0), (W // 2, H // 2), (W - 1, H - 1), (W // 3, 2 * H // 3), (5, H - 7)]
        sv["camera_rays"] = [{"x": x, "y": y, "iter": 3, "ray": f32list(o.camera_ray(x, y, 3))} for x, y in px]
        hits = []
        for x, y in px:
            r = o.camera_ray(x, y, 3)
            pid, h = o.trace(r[:3], r[3:], 1e-3)
            hits.append({"origin": f32list(r[:3]), "dir": f32list(r[3:]), "prim": int(pid), "hit": f32list(h),
                         "occluded_to_5": bool(o.occluded(r[:3], r[3:], 1e-3, 5.0))})
        sv["trace"] = hits
        bs = []
        n = np.float32([0.0, 1.0, 0.0])
        dpdu = np.float32([1.0, 0.0, 0.0])
        wos = [np.float32([0.3, 0.8, 0.2]), np.float32([-0.5, 0.4, 0.6]), np.float32([0.2, -0.7, 0.1])]
        us = [np.float32([0.13, 0.71, 0.42]), np.float32([0.83, 0.27, 0.91]), np.float32([0.5, 0.5, 0.05])]
        for m in range(o.n_materials):
            for wo in wos:
                wo = wo / np.linalg.norm(wo)
                for u in us:
                    s = o.bsdf_sample(m, wo, n, dpdu, [0.3, 0.6], u)
                    e = o.bsdf_eval(m, wo, s[:3], n, dpdu, [0.3, 0.6])
                    bs.append({"mat": m, "wo": f32list(wo), "u": f32list(u), "sample": f32list(s), "eval": f32list(e)})
        sv["bsdf"] = bs
        sv["lights"] = [{"idx": i, "sample": f32list(o.light_sample(i, [0.1, 0.5, 0.2], [0.37, 0.58]))}
                        for i in range(min(o.n_lights, 4))]
        sv["samples"] = [{"pixel": int(p), "iter": it, "ok": bool(o.sample(int(p), it)[0]),
                          "rgb": f32list(o.sample(int(p), it)[1])}
                         for p in (0, W * H // 2 + W // 2, W * H - 1) for it in (1, 2)]
        vec["scenes"][key] = sv
        print(key, "mean", acc.mean() / SPP)
    with open(os.path.join(HERE, "vectors.json"), "w") as f:
        json.dump(vec, f, indent=0)
    make_light_tracing()


def make_light_tracing():
    out = {}
    for key, kw in LT_SCENES.items():
        kw = dict(kw)
        name = kw.pop("name")
        path = gen_scenes.generate(os.path.join(HERE, "scenes", key), name, **kw)
        o = ol.Oracle(path)
        c = ol.Counts()
        acc = o.render(1, LT_SPP, LT_THREADS, counts=c)
        np.save(os.path.join(HERE, key + "_sum.npy"), acc)
        np.save(os.path.join(HERE, key + "_ids.npy"), o.primary_ids(1))
        out[key] = {"spp": LT_SPP, "threads": LT_THREADS, "counts": c.as_dict()}
        print(key, "mean", acc.mean() / LT_SPP)
    with open(os.path.join(HERE, "light_tracing.json"), "w") as f:
        json.dump(out, f, indent=0)


if __name__ == "__main__":
    if sys.argv[1:] == ["lt"]:   # only the light-tracing fixtures
        make_light_tracing()
    else:
        main()
