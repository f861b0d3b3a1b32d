#!/usr/bin/env python3
"""How much the order of the rays in a queue is worth to the traversal kernel (experiment, GPU).

  python tools/exp/ray_order.py [--scene car_pt] [--width 1920 --height 1080] [--spp 4]

Builds the rays of bounce 1, 2 and 3 of a diffuse path from the library's own hooks (camera rays, hit records,
cosine-distributed directions about the shading normal) and times the traversal launch of caster_trace_rays
(caster_get_kernel_times) over the same set of rays in four orders:
  queue    the order the renderer's queues have: pixel after pixel (origins of neighbours lie together, directions do not)
  octant   stable partition by the signs of the direction
  sorted   by octant, then by the Morton code of the origin (10 bits per axis)
  shuffled a random permutation (the worst case)
"""
import argparse
import importlib
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def morton(q):
    def spread(v):
        v = v.astype(np.uint64) & 0x3ff
        v = (v | (v << 16)) & 0x30000ff
        v = (v | (v << 8)) & 0x300f00f
        v = (v | (v << 4)) & 0x30c30c3
        v = (v | (v << 2)) & 0x9249249
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)


def cosine_dirs(rng, n):
    u1, u2 = rng.uniform(size=len(n)), rng.uniform(size=len(n))
    r, phi = np.sqrt(u1), 2 * np.pi * u2
    local = np.stack([r * np.cos(phi), r * np.sin(phi), np.sqrt(np.maximum(0, 1 - u1))], 1)
    a = np.where(np.abs(n[:, :1]) > 0.9, np.array([[0, 1, 0]]), np.array([[1, 0, 0]]))
    t = np.cross(a, n); t /= np.linalg.norm(t, axis=1, keepdims=True)
    b = np.cross(n, t)
    return (local[:, :1] * t + local[:, 1:2] * b + local[:, 2:] * n).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="car_pt")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=4)
    a = ap.parse_args()
    import gen_scenes
    caster = importlib.import_module("cuda-pathtracer_amd")
    d = tempfile.mkdtemp()
    path = gen_scenes.generate(d, a.scene, 1.0, a.width, a.height, 8)
    sc = caster.Scene(path)
    r = caster.Renderer(sc, 0, a.spp, 1)
    r.set_profiling(True)
    rng = np.random.default_rng(3)
    ys, xs = np.mgrid[0:a.height, 0:a.width]
    rows = []
    for it in range(1, a.spp + 1):   # iteration-major, pixel after pixel: the order of a batch's slots
        rows.append(np.stack([xs.reshape(-1), ys.reshape(-1), np.full(xs.size, it)], 1))
    rows = np.concatenate(rows).astype(np.int32)
    cam = r.eval_function("camera_ray", rows.view(np.float32))
    o, dr = cam[:, :3].copy(), cam[:, 3:6].copy()

    def timed(o, dr, label):
        tm = np.full(len(o), np.inf, np.float32)
        r.trace_rays(o[:1000], dr[:1000], tm[:1000])   # (warm)
        best = 1e9
        for _ in range(3):
            before = r.kernel_times()["extend"][0]
            prim, t = r.trace_rays(o, dr, tm)
            best = min(best, r.kernel_times()["extend"][0] - before)
        print("    %-9s %8.3f ms  %7.1f Mrays/s" % (label, best, len(o) / best / 1e3), flush=True)
        return prim, t

    for bounce in (1, 2, 3):
        n = len(o)
        print("bounce %d: %d rays" % (bounce, n), flush=True)
        prim, t = timed(o, dr, "queue")
        if bounce > 1:
            octant = (dr[:, 0] < 0).astype(np.int64) | ((dr[:, 1] < 0).astype(np.int64) << 1) | ((dr[:, 2] < 0).astype(np.int64) << 2)
            k = np.argsort(octant, kind="stable")
            timed(o[k], dr[k], "octant")
            lo, hi = o.min(0), o.max(0)
            q = np.clip((o - lo) / np.maximum(hi - lo, 1e-9) * 1023, 0, 1023).astype(np.int64)
            k = np.lexsort((morton(q), octant))
            timed(o[k], dr[k], "sorted")
            k = np.argsort(morton(q), kind="stable")
            timed(o[k], dr[k], "morton")
            k = rng.permutation(n)
            timed(o[k], dr[k], "shuffled")
        hit = prim >= 0
        rec_in = np.concatenate([o, dr, t[:, None], prim.view(np.float32)[:, None]], 1)[hit]
        rec = r.eval_function("hit_record", rec_in)
        pos, nrm = rec[:, 0:3], rec[:, 3:6].astype(np.float64)
        din = dr[hit]
        nrm = np.where((nrm * din).sum(1, keepdims=True) > 0, -nrm, nrm)   # towards the side the ray came from
        dr = cosine_dirs(rng, nrm)
        o = (pos + 1e-3 * nrm).astype(np.float32)
    r.close()


if __name__ == "__main__":
    main()
