"""The BVH built on the GPU (caster_scene_load_device) against the host build
(caster_scene_load): same nodes in the same pre-order, same leaf ranges, same
primitive order.  Boxes are compared with ==, which holds for every value except
that a zero may differ in sign (minimum/maximum of +0 and -0 is either)."""
import time

import numpy as np
import pytest

import gen_scenes

pytestmark = pytest.mark.gpu


def compare(caster, path):
    t0 = time.perf_counter()
    host = caster.Scene(path)
    t1 = time.perf_counter()
    dev = caster.Scene(path, build_device=0)
    t2 = time.perf_counter()
    assert host.info.n_nodes == dev.info.n_nodes and host.info.n_prims == dev.info.n_prims
    hb, hm, hi = host.bvh()
    db, dm, di = dev.bvh()
    np.testing.assert_array_equal(hm, dm)
    np.testing.assert_array_equal(hi, di)
    assert (hb == db).all()
    print("%s: %d primitives, %d nodes; load with host build %.2f s, with GPU build %.2f s" % (
        path.split("/")[-2], host.info.n_prims, host.info.n_nodes, t1 - t0, t2 - t1))
    return dev


@pytest.mark.parametrize("key", ["cornell_pt", "cornell_vpt", "stairs_pt", "knot_pt", "car_env"])
def test_small_scenes(caster, scenes, key):
    compare(caster, scenes[key])


def test_single_primitive_and_tiny_scenes(caster, tmp_path):
    import json
    for k, shapes in enumerate(([{"sphere": True, "material": "m"}],
                                [{"sphere": True, "center": [i, 0, 0], "radius": 0.3, "material": "m"} for i in range(5)],
                                [{"sphere": True, "center": [0, 0, 0], "radius": 0.3, "material": "m"}] * 9)):
        doc = {"camera": {}, "material": [{"name": "m", "bsdf": "lambertian"}], "scene": shapes}
        p = tmp_path / ("s%d.json" % k)
        p.write_text(json.dumps(doc))
        compare(caster, str(p))


def test_full_size_scene_renders_the_same(caster, tmp_path):
    path = gen_scenes.generate(str(tmp_path / "car"), "car_pt", 1.0, 1920, 1080, 8)
    dev = compare(caster, path)
    host = caster.Scene(path)
    a = caster.Renderer(host, 0, 1, 1)
    b = caster.Renderer(dev, 0, 1, 1)
    np.testing.assert_array_equal(a.primary_ids(1), b.primary_ids(1))
    a.render(1, 1, True)
    b.render(1, 1, True)
    np.testing.assert_array_equal(a.read_film(2), b.read_film(2))
    a.close()
    b.close()
