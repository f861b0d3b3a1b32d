#!/usr/bin/env python3
"""Narrows a pixel whose value changes between repeats down to the iteration and the path depth."""
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
pixels = [(168, 343), (356, 665), (736, 671)]
tmp = tempfile.mkdtemp()
for depth in (8, 1, 2, 3, 4):
    path = gen_scenes.generate(os.path.join(tmp, "d%d" % depth), "stairs_pt", 0.3, 1920, 1080, depth)
    sc = caster.Scene(path)
    o = ol.Oracle(path) if depth == 8 else None
    r = caster.Renderer(sc, 0, 24)
    for (y, x) in pixels:
        for it in range(1, 25):
            vals = set()
            for rep in range(6):
                r.render(it, 1, True)
                vals.add(tuple(r.read_film(2)[y, x].tolist()))
            if len(vals) > 1:
                msg = "depth %d pixel %s iteration %d: %s" % (depth, (y, x), it, sorted(vals))
                if o is not None:
                    msg += "  oracle %s" % (o.render(it, 1, 1, rows=(y, y + 1))[y, x].tolist(),)
                print(msg, flush=True)
    r.close()
