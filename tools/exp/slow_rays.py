#!/usr/bin/env python3
"""Diagnosis: the rays a -DCASTER_DEBUG_SLOW_RAYS=N build of the library recorded as taking more than N steps.
  CASTER_LIB=.../slow.so python tools/exp/slow_rays.py [scene] [first_iter n_iter]"""
import ctypes as C
import importlib
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_scenes
caster = importlib.import_module("cuda-pathtracer_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "stairs_pt"
first, n = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (65, 64)
d = "/tmp/slow_rays_%s" % name
path = os.path.join(d, "scene.json")
if not os.path.exists(path):
    gen_scenes.generate(d, name, 1.0, 1920, 1080, 8)
sc = caster.Scene(path)
r = caster.Renderer(sc, 0, 64, 1)
r.render(first, n, True)
out = np.zeros((256, 12), np.float32)
cnt = C.c_uint(0)
L = caster.lib()
L.caster_debug_slow_rays.argtypes = [C.c_void_p, C.POINTER(C.c_uint)]
assert L.caster_debug_slow_rays(out.ctypes.data, C.byref(cnt)) == 0
print("slow rays:", cnt.value)
np.set_printoptions(precision=9, linewidth=200, suppress=False)
for row in out[:min(cnt.value, 24)]:
    print("o", row[0:3], "seed", row[3:4].view(np.int32)[0], "d", row[4:7], "tmax", row[7], "steps", int(row[8]), "any" if row[9] else "closest",
          "hit t", row[10], "prim", row[11:12].view(np.int32)[0])
r.close()
