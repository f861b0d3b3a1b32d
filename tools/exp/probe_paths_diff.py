#!/usr/bin/env python3
"""Films of the two ways of tracing probe rays (libraries built with CASTER_PROBE_ANY_MAX_SLOTS = 0 and
0xffffffff under cuda-pathtracer_amd/variants/) at two batch sizes, against each other and, on the rows
where they differ, against the CPU oracle.  Each library runs in a process of its own."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
SPP = 24

if len(sys.argv) > 1:   # child: render with the library CASTER_LIB names
    import importlib
    caster = importlib.import_module("cuda-pathtracer_amd")
    sc = caster.Scene(sys.argv[1])
    for k in (2, 24):
        r = caster.Renderer(sc, 0, k)
        r.render(1, SPP, True)
        np.save(sys.argv[2] + "_%d.npy" % k, r.read_film(2))
        print(sys.argv[2], k, r.stats()["rays"], "overflow", r.stats()["stack_overflow"], flush=True)
        r.close()
    sys.exit(0)

import gen_scenes
import oracle_lib as ol
tmp = tempfile.mkdtemp()
path = gen_scenes.generate(tmp, "stairs_pt", 0.3, 1920, 1080, 8)
films = {}
for lib in ("never", "always"):
    env = dict(os.environ, CASTER_LIB=os.path.join(ROOT, "cuda-pathtracer_amd", "variants", lib + ".so"))
    subprocess.run([sys.executable, __file__, path, os.path.join(tmp, lib)], env=env, check=True, timeout=500)
    for k in (2, 24):
        films[(lib, k)] = np.load(os.path.join(tmp, lib) + "_%d.npy" % k)
ref = films[("always", 2)]
rows = set()
for key, f in films.items():
    bad = np.argwhere((f != ref).any(axis=2))
    print(key, len(bad), "pixels differ from (always, 2)", bad[:6].tolist())
    rows |= set(int(b[0]) for b in bad)
o = ol.Oracle(path)
for y in sorted(rows)[:4]:
    cpu = o.render(1, SPP, 8, rows=(y, y + 1))[y]
    for key, f in films.items():
        print("row", y, key, "pixels unequal to the oracle:", int((f[y] != cpu).any(axis=1).sum()))
