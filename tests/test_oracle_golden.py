"""The CPU oracle against the committed golden fixtures (tests/golden), the
Thrust known answer, and a BVH-free brute force."""
import json
import os

import numpy as np
import pytest

import oracle_lib as ol

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
VEC = json.load(open(os.path.join(GOLD, "vectors.json")))
SCENES = sorted(VEC["scenes"])


def scene_path(key):
    return os.path.join(GOLD, "scenes", key, "scene.json")


def test_minstd_known_answer():
    # thrust/random/linear_congruential_engine.h: "the 10000th consecutive invocation of a
    # default-constructed object of type minstd_rand shall produce the value 399268537"
    assert int(ol.rng_raw(1, 10000)[-1]) == 399268537
    assert VEC["rng"]["minstd_10000th_seed1"] == 399268537


def test_rng_stream_and_hash():
    for k, v in VEC["rng"]["wang_hash"].items():
        assert ol.wang_hash(int(k)) == v
    seed = VEC["rng"]["seed_pixel0_iter1"]
    assert (ol.wang_hash(0) + ol.wang_hash(1)) & 0xffffffff == seed
    assert [int(x) for x in ol.rng_raw(seed, 8)] == VEC["rng"]["first8_raw_pixel0_iter1"]
    got = ol.rng_floats(seed, 8)
    assert got.tolist() == VEC["rng"]["first8_floats_pixel0_iter1"]
    # u = float(x - 1) / 2^31 stays inside [0, 1]
    u = ol.rng_floats(12345, 100000)
    assert u.min() >= 0.0 and u.max() <= 1.0


@pytest.mark.parametrize("key", SCENES)
def test_function_vectors(key):
    sv = VEC["scenes"][key]
    o = ol.Oracle(scene_path(key))
    assert (o.n_prims, o.n_nodes, o.n_lights) == (sv["n_prims"], sv["n_nodes"], sv["n_lights"])
    for c in sv["camera_rays"]:
        assert o.camera_ray(c["x"], c["y"], c["iter"]).tolist() == c["ray"]
    for t in sv["trace"]:
        pid, h = o.trace(t["origin"], t["dir"], 1e-3)
        assert pid == t["prim"]
        if pid >= 0:
            assert h.tolist() == t["hit"]
        assert o.occluded(t["origin"], t["dir"], 1e-3, 5.0) == t["occluded_to_5"]
    n, dpdu = [0.0, 1.0, 0.0], [1.0, 0.0, 0.0]
    for b in sv["bsdf"]:
        s = o.bsdf_sample(b["mat"], b["wo"], n, dpdu, [0.3, 0.6], b["u"])
        np.testing.assert_array_equal(s, np.float32(b["sample"]))
        e = o.bsdf_eval(b["mat"], b["wo"], s[:3], n, dpdu, [0.3, 0.6])
        np.testing.assert_array_equal(e, np.float32(b["eval"]))
    for l in sv["lights"]:
        np.testing.assert_array_equal(o.light_sample(l["idx"], [0.1, 0.5, 0.2], [0.37, 0.58]), np.float32(l["sample"]))
    for s in sv["samples"]:
        ok, rgb = o.sample(s["pixel"], s["iter"])
        assert ok == s["ok"]
        np.testing.assert_array_equal(rgb, np.float32(s["rgb"]))


@pytest.mark.parametrize("key", SCENES)
def test_images_and_ids(key):
    sv = VEC["scenes"][key]
    o = ol.Oracle(scene_path(key))
    c = ol.Counts()
    acc = o.render(1, sv["spp"], 4, counts=c)
    np.testing.assert_array_equal(acc, np.load(os.path.join(GOLD, key + "_sum.npy")))
    np.testing.assert_array_equal(o.primary_ids(1), np.load(os.path.join(GOLD, key + "_ids.npy")))
    assert c.as_dict() == sv["counts"]
    # thread count and row splitting do not change a single bit
    acc1 = o.render(1, sv["spp"], 1)
    np.testing.assert_array_equal(acc, acc1)
    half = o.height // 2
    acc2 = o.render(1, sv["spp"], 2, rows=(0, half))
    o.render(1, sv["spp"], 2, acc=acc2, rows=(half, o.height))
    np.testing.assert_array_equal(acc, acc2)
    # accumulation is in iteration order: two halves chained equal one run
    acc3 = o.render(1, sv["spp"] // 2, 4)
    o.render(1 + sv["spp"] // 2, sv["spp"] - sv["spp"] // 2, 4, acc=acc3)
    np.testing.assert_array_equal(acc, acc3)


SPLAT = json.load(open(os.path.join(GOLD, "light_tracing.json")))


@pytest.mark.parametrize("key", sorted(SPLAT))
def test_splatting_integrators(key):
    """Light tracing and BDPT add contributions to pixels other than the sample's own, so the
    oracle's film depends on its thread count in the rounding of the per-thread buffers' sum:
    the fixture is reproduced bit for bit at the thread count it was made with."""
    sv = SPLAT[key]
    o = ol.Oracle(scene_path(key))
    c = ol.Counts()
    acc = o.render(1, sv["spp"], sv["threads"], counts=c)
    np.testing.assert_array_equal(acc, np.load(os.path.join(GOLD, key + "_sum.npy")))
    np.testing.assert_array_equal(o.primary_ids(1), np.load(os.path.join(GOLD, key + "_ids.npy")))
    assert c.as_dict() == sv["counts"]
    # another thread count adds the same numbers in another order
    acc1 = o.render(1, sv["spp"], 1)
    d = (acc1 - acc) / (np.abs(acc) + 1e-2)
    assert float(np.sqrt(np.mean(d * d))) <= 1e-6
    # energy: the film of a splatting integrator is a sum over all samples wherever they land
    assert acc.sum() > 0 and np.isfinite(acc).all()


@pytest.mark.parametrize("key", ["cornell_ao", "cornell_vpt", "stairs_pt"])
def test_bvh_matches_bruteforce(key):
    """Closest hits found through the BVH equal a linear scan over every primitive."""
    o = ol.Oracle(scene_path(key))
    for it in (1, 5):
        np.testing.assert_array_equal(o.primary_ids(it), o.primary_ids(it, bruteforce=True))


def test_ao_sample_values():
    """On a triangle (orthonormal shading frame) an AO sample is 0 when occluded and
    cos/pdf = pi when not; a miss is 1.  (Spheres carry the reference's non-orthogonal
    tangent, sphere.h:81, so their values spread.)"""
    o = ol.Oracle(scene_path("cornell_ao"))
    ids = o.primary_ids(1)
    seen = set()
    for p in range(0, o.width * o.height, 11):
        pid = ids.ravel()[p]
        ok, rgb = o.sample(p, 1)
        assert ok and rgb[0] == rgb[1] == rgb[2]
        if pid < 0:
            assert rgb[0] == 1.0
        elif pid not in (0, 1) or True:
            seen.add(round(float(rgb[0]), 3))
    assert 0.0 in seen and 3.142 in seen


def test_tonemap_matches_formula():
    o = ol.Oracle(scene_path("cornell_pt"))
    x = np.float32([[0.0, 0.004, 0.1], [1.0, 5.0, 100.0]])
    f = o.tonemap(x, True)
    c = np.maximum(x.astype(np.float64) - 0.004, 0)
    ref = (c * (6.2 * c + 0.5)) / (c * (6.2 * c + 1.7) + 0.06)
    np.testing.assert_allclose(f, ref, rtol=1e-5, atol=1e-7)
    g = o.tonemap(x, False)
    ref = np.power(np.maximum(x.astype(np.float64), 1e-5) * 1.41421356, 1 / 2.2)
    np.testing.assert_allclose(g, ref, rtol=1e-5)


def test_sample_camera_inverts_primary_rays():
    """Camera::SampleCamera (Camera.h:87-118) against GeneratePrimaryRay (:45-85): a point on the
    primary ray of pixel (x, y) is seen by the pinhole at that pixel (the raster position is rounded
    to a grid of res - 1 steps, so the neighbour is allowed), with density dist^2 / cos and
    importance d^2 / (A cos^4), A = 4 * half width * half height of the image plane at distance
    d = 0.1 (parsescene.cpp:608); a point behind the camera is not seen."""
    path = scene_path("cornell_fog_lt")
    o = ol.Oracle(path)
    W, H = o.width, o.height
    fov = json.load(open(path))["camera"].get("fov", 60.0)
    d = 0.1
    half_h = np.tan(np.radians(fov * 0.5)) * d
    area = 4.0 * (half_h * W / H) * half_h
    rng = np.random.RandomState(3)
    for _ in range(200):
        x, y = int(rng.randint(1, W - 1)), int(rng.randint(1, H - 1))
        ray = o.camera_ray(x, y, 1 + int(rng.randint(50))).astype(np.float64)
        org, dr = ray[:3], ray[3:]
        t = rng.uniform(0.5, 6.0)
        raster, (we, pdf, tmax) = o.sample_camera(org + t * dr)
        assert raster >= 0
        rx, ry = raster % W, raster // W
        assert abs(rx - x) <= 1 and abs(ry - y) <= 1
        assert abs(tmax - (t - 1e-3)) < 1e-3
        cos = t * t / pdf                      # pdf = dist^2 / cos
        assert 0 < cos <= 1 + 1e-5
        assert abs(we * area * cos ** 4 / (d * d) - 1) < 1e-3
    centre = o.camera_ray(W // 2, H // 2, 1).astype(np.float64)
    raster, (we, pdf, tmax) = o.sample_camera(centre[:3] - 2.0 * centre[3:])
    assert raster == -1 and pdf == 0


def test_bidirectional_and_unidirectional_estimators_agree(tmp_path):
    """Path tracing and BDPT estimate the same image (BDPT without the depth limit, so slightly
    brighter): block means of the Cornell box agree within noise.  Guards the MIS weights, which
    no bit-exact comparison between two implementations of the same formulas can."""
    import gen_scenes
    films = {}
    for name in ("cornell_pt", "cornell_bdpt"):
        path = gen_scenes.generate(str(tmp_path / name), name, width=32, height=32, depth=12)
        films[name] = ol.Oracle(path).render(1, 192, 8) / 192.0
    pt = films["cornell_pt"].reshape(4, 8, 4, 8, 3).mean(axis=(1, 3))
    bd = films["cornell_bdpt"].reshape(4, 8, 4, 8, 3).mean(axis=(1, 3))
    ratio = bd.sum() / pt.sum()
    assert 0.97 < ratio < 1.08, ratio
    rel = np.abs(bd - pt).sum(axis=2) / (pt.sum(axis=2) + 1e-3)
    assert np.median(rel) < 0.1, rel


# Size of a declared deviation.  The reference's sphere test (sphere.h:24-66) assumes a unit direction; for the rays of
# other lengths that leave a line or a sphere its roots are no distances, and a candidate in front of the point where
# the ray enters the sphere's own bounding box is rejected here (oracle and renderer alike: the closest hit must not
# depend on the order the tree is walked in).  The oracle counts the candidates the reference would have accepted and
# this rule drops: a few in 10^4 samples where spheres are seen through other spheres or through a medium, none where
# every ray has unit length.
SPHERE_BOX_REJECTS = {"cornell_ao": 6, "cornell_pt": 0, "cornell_vpt": 57, "stairs_pt": 4, "car_env": 0}


@pytest.mark.parametrize("key", sorted(SPHERE_BOX_REJECTS))
def test_size_of_the_sphere_deviation(key):
    o = ol.Oracle(os.path.join(GOLD, "scenes", key, "scene.json"))
    c = ol.Counts()
    o.render(1, 8, 4, counts=c)
    d = c.as_dict()
    print(key, "samples", d["samples"], "sphere tests", sum(d["spheres"]), "dropped by the rule", d["sphere_box_rejects"])
    assert d["sphere_box_rejects"] == SPHERE_BOX_REJECTS[key]
    assert d["sphere_box_rejects"] <= 2e-3 * d["samples"]
