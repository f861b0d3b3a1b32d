#!/usr/bin/env python3
"""Rays and traversal counters (node records fetched, primitive records tested, per ray kind) of a short render.
  CASTER_LIB=... python tools/exp/walk_counts.py [scene] [width height]"""
import importlib
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_scenes
caster = importlib.import_module("cuda-pathtracer_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "stairs_pt"
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (960, 540)
d = "/tmp/walk_counts_%s_%d" % (name, w)
path = os.path.join(d, "scene.json")
if not os.path.exists(path):
    gen_scenes.generate(d, name, 1.0, w, h, 8)
sc = caster.Scene(path)
r = caster.Renderer(sc, 0, 8, 1)
r.render(1, 8, True)
st, wc = r.stats(), r.walk_counts()
print(os.environ.get("CASTER_LIB", "libcaster.so")[-30:], "rays", st["rays"], "invalid", st["invalid"])
for kind, rays in (("closest", st["rays"][0] + st["rays"][2]), ("any", st["rays"][1] + (st["rays"][2] if False else 0))):
    print("  %-8s nodes/ray %.2f prims/ray %.2f" % (kind, wc[kind]["nodes"] / max(rays, 1), wc[kind]["prims"] / max(rays, 1)))
print("  film mean", float(r.read_film(0).mean()))
r.close()
