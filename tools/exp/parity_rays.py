#!/usr/bin/env python3
"""Every ray of one oracle sample (pixel, iteration), traced as a closest-hit ray by the GPU's traversal
kernel (caster_trace_rays) and by the oracle (oracle_trace): which ray do the two sides answer differently?

  parity_rays.py SCENE DETAIL Y X ITERATION [MAXDEPTH]
"""
import ctypes as C
import importlib
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen_scenes
import oracle_lib as ol

caster = importlib.import_module("cuda-pathtracer_amd")
name, detail, y, x, it = sys.argv[1], float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
depth = int(sys.argv[6]) if len(sys.argv) > 6 else 8
tmp = tempfile.mkdtemp()
path = gen_scenes.generate(tmp, name, detail, 1920, 1080, depth)
o = ol.Oracle(path)
log = os.path.join(tmp, "rays.txt")
sys.stderr.flush()
saved = os.dup(2)
fd = os.open(log, os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
os.dup2(fd, 2)
o.L.oracle_debug_bruteforce(1)
rgb = (C.c_float * 3)()
o.L.oracle_sample(o.h, C.c_uint32(y * 1920 + x), C.c_uint32(it), rgb)
o.L.oracle_debug_bruteforce(0)
os.dup2(saved, 2)
os.close(fd)
rays = [l.split()[1:] for l in open(log) if l.startswith("RAY ")]
print("oracle sample", list(rgb), "-", len(rays), "rays")
kinds = [int(r[0]) for r in rays]
vals = np.array([[float(v) for v in r[1:]] for r in rays], np.float32)
sc = caster.Scene(path)
r = caster.Renderer(sc, 0, 1)
prim, t = r.trace_rays(vals[:, 0:3], vals[:, 3:6], vals[:, 6])
r.close()
o.L.oracle_trace.restype = C.c_int
eps = sc.info.epsilon
for i in range(len(rays)):
    out = (C.c_float * 12)()
    org = (C.c_float * 3)(*vals[i, 0:3])
    d = (C.c_float * 3)(*vals[i, 3:6])
    p = o.L.oracle_trace(o.h, org, d, C.c_float(eps), C.c_float(vals[i, 6]), out)
    same = p == prim[i] and (p < 0 or out[0] == t[i])
    print("ray %d kind %d tmax %.9g: oracle prim %d t %.9g | gpu prim %d t %.9g %s" % (
        i, kinds[i], vals[i, 6], p, out[0], prim[i], t[i], "" if same else "  <-- differ"))
