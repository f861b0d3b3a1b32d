#!/usr/bin/env python3
"""Per-configuration visit counts of the reference's own traversal order, counted by the CPU
oracle: rays, BVH nodes visited and primitives tested per ray class (extension, shadow, MIS
probe, transmittance) on a bounded sample of each benchmark configuration at its full
resolution (1 sample per pixel on every 8th block of 8 scanlines).  These are the counts the
algorithmic bytes of bench.py's roofline are built from; bench.py recounts them live for the
workload it times, this file pins them for all five configurations.

usage: make_counts.py [OUT.json]   (default tests/golden/counts.json)
"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen_scenes  # noqa: E402
import oracle_lib as ol  # noqa: E402

CONFIGS = [
    ("C1 cornell_ao 512x512 ao", "cornell_ao", 512, 512, 0),
    ("C2 knot_pt 1024x1024 pt maxDepth 8", "knot_pt", 1024, 1024, 8),
    ("C3 stairs_pt 1920x1080 pt maxDepth 8", "stairs_pt", 1920, 1080, 8),
    ("C4 cornell_vpt 1024x1024 vpt maxDepth 16", "cornell_vpt", 1024, 1024, 16),
    ("C5 car_pt 1920x1080 pt maxDepth 8", "car_pt", 1920, 1080, 8),
]


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "counts.json")
    out = {"sample": "1 spp on every 8th block of 8 scanlines, iteration 1", "classes": ["extension", "shadow", "probe", "transmittance"],
           "bytes": {"node": 32, "triangle": 36, "sphere": 16, "line": 32}, "configs": {}}
    with tempfile.TemporaryDirectory() as tmp:
        for label, name, w, h, depth in CONFIGS:
            path = gen_scenes.generate(os.path.join(tmp, name), name, 1.0, w, h, depth)
            o = ol.Oracle(path)
            c = ol.Counts()
            for y in range(0, h, 64):
                o.render(1, 1, os.cpu_count() or 1, counts=c, rows=(y, min(y + 8, h)))
            d = c.as_dict()
            per_ray = []
            for k in range(4):
                r = d["rays"][k]
                per_ray.append(0.0 if r == 0 else (32.0 * d["nodes"][k] + 36.0 * d["tris"][k] + 16.0 * d["spheres"][k]
                                                   + 32.0 * d["lines"][k]) / r)
            out["configs"][label] = {"counts": d, "traversal_bytes_per_ray": per_ray,
                                     "rays_per_sample": sum(d["rays"]) / max(d["samples"], 1)}
            print(label, "rays/sample %.2f" % out["configs"][label]["rays_per_sample"],
                  "bytes/ray", [round(x) for x in per_ray], flush=True)
    with open(out_path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
