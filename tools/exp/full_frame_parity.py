#!/usr/bin/env python3
"""The whole 1920x1080 frame at N samples per pixel on the GPU against the CPU oracle, pixel by pixel
(the tests compare bands of a few scanlines; this is the same comparison over tens of millions of samples)."""
import importlib
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
name = sys.argv[1] if len(sys.argv) > 1 else "stairs_pt"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
detail = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
threads = int(sys.argv[4]) if len(sys.argv) > 4 else 16
path = gen_scenes.generate(tempfile.mkdtemp(), name, detail, 1920, 1080, 8)
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
for y0 in range(0, 1080, 120):          # in bands, so that a long run keeps printing
    cpu[y0:y0 + 120] = o.render(1, spp, threads, rows=(y0, y0 + 120), counts=counts)[y0:y0 + 120]
    print("oracle rows", y0 + 120, "after %.0f s" % (time.time() - t0), flush=True)
bad = np.argwhere((gpu != cpu).any(axis=2))
c = counts.as_dict()
print("%s: %d samples, %d pixels differ %s; rays gpu %s oracle %s" % (
    name, 1920 * 1080 * spp, len(bad), bad[:6].tolist(), st["rays"], c["rays"]))
