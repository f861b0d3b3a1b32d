#!/usr/bin/env python3
"""A whole frame of light tracing or BDPT on the GPU against the CPU oracle: relative RMSE per pixel, pixels beyond 1e-4,
ray counts (the films are sums of the same numbers in another order).

  python tools/exp/full_frame_splat_parity.py SCENE INTEGRATOR SPP [WIDTH HEIGHT] [THREADS]"""
import importlib
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen_scenes
import oracle_lib as ol

caster = importlib.import_module("cuda-pathtracer_amd")
name, integ, spp = sys.argv[1], sys.argv[2], int(sys.argv[3])
w = int(sys.argv[4]) if len(sys.argv) > 4 else 1920
h = int(sys.argv[5]) if len(sys.argv) > 5 else 1080
threads = int(sys.argv[6]) if len(sys.argv) > 6 else 16
path = gen_scenes.generate(tempfile.mkdtemp(), name, 1.0, w, h, 8)
doc = json.load(open(path))
doc["integrator"] = integ
json.dump(doc, open(path, "w"))
sc = caster.Scene(path)
r = caster.Renderer(sc, 0, 0)
r.render(1, spp, True)
gpu = r.read_film(2)
st = r.stats()
r.close()
o = ol.Oracle(path)
counts = ol.Counts()
t0 = time.time()
cpu = np.zeros_like(gpu)
for it in range(1, spp + 1):            # an iteration at a time, so that a long run keeps printing
    cpu += o.render(it, 1, threads, counts=counts)
    print("oracle iteration", it, "after %.0f s" % (time.time() - t0), flush=True)
d = (gpu - cpu) / (np.abs(cpu) + 1e-2)
c = counts.as_dict()
print("%s %s: %d samples, relRMSE %.3g, pixels beyond 1e-4: %d of %d, max %.3g; rays gpu %s oracle %s" % (
    name, integ, w * h * spp, float(np.sqrt(np.mean(d * d))), int((np.abs(d) > 1e-4).any(axis=2).sum()), w * h,
    float(np.abs(d).max()), [st["rays"][0], st["rays"][3]], [c["rays"][0], c["rays"][3]]))
