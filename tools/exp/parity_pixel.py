#!/usr/bin/env python3
"""For pixels where the GPU film differs from the oracle's: the iteration, the two values, and the path
depth (maxDepth of the scene file) from which on they differ."""
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
name, spp, detail = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
pixels = [tuple(int(v) for v in a.split(",")) for a in sys.argv[4:]]
tmp = tempfile.mkdtemp()
found = {}
for depth in (8, 1, 2, 3, 4, 5):
    path = gen_scenes.generate(os.path.join(tmp, "d%d" % depth), name, detail, 1920, 1080, depth)
    sc = caster.Scene(path)
    o = ol.Oracle(path)
    r = caster.Renderer(sc, 0, 1)
    for (y, x) in pixels:
        r.set_rows(y - y % 8, y - y % 8 + 8)
        its = range(1, spp + 1) if depth == 8 else found.get((y, x), [])
        for it in its:
            r.render(it, 1, True)
            g = r.read_film(2)[y, x]
            c = o.render(it, 1, 1, rows=(y, y + 1))[y, x]
            if (g != c).any():
                if depth == 8:
                    found.setdefault((y, x), []).append(it)
                print("maxDepth %d pixel %s iteration %d: gpu %s oracle %s" % (depth, (y, x), it, g.tolist(), c.tolist()), flush=True)
    r.close()
