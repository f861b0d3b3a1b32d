#!/usr/bin/env python3
"""Renders the same frame several times with the same settings and counts the pixels that differ between
repeats (the film is a pure function of scene, pixel and iteration: there should be none)."""
import importlib
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_scenes

caster = importlib.import_module("cuda-pathtracer_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "stairs_pt"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 24
path = gen_scenes.generate(tempfile.mkdtemp(), name, 0.3, 1920, 1080, 8)
sc = caster.Scene(path)
for k, streams in ((2, 0), (24, 0), (24, 1)):
    films = []
    for rep in range(3):
        r = caster.Renderer(sc, 0, k, streams) if streams else caster.Renderer(sc, 0, k)
        r.render(1, spp, True)
        films.append(r.read_film(2))
        r.close()
    for rep in (1, 2):
        bad = np.argwhere((films[rep] != films[0]).any(axis=2))
        print(name, "k", k, "streams", streams or "default", "repeat", rep, ":", len(bad), "pixels differ", bad[:4].tolist(), flush=True)
