"""The device functions of the shade stage, one by one, against the CPU oracle's.

In a film the BSDF, light and camera code is only seen through whole paths; caster_eval_function runs each
function on inputs of the caller's.  Here: (1) the per-function fixtures of tests/golden/vectors.json, which the
oracle tests run on the CPU, through the HIP functions; (2) thousands of random inputs per material, emitter and
pixel through both sides.  Same bits on both sides (a NaN counts as equal to a NaN)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
VEC = json.load(open(os.path.join(GOLD, "vectors.json")))
SCENES = sorted(VEC["scenes"])


def bits(i):
    return np.array([i], np.int32).view(np.float32)[0]


def same(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


def unit(rng, n):
    v = rng.normal(size=(n, 3))
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


@pytest.fixture(scope="module")
def pairs(caster):
    out = {}
    for key in SCENES:
        path = os.path.join(GOLD, "scenes", key, "scene.json")
        sc = caster.Scene(path)
        out[key] = (sc, caster.Renderer(sc, 0, 1, 1), ol.Oracle(path))
    yield out
    for sc, r, o in out.values():
        r.close()


@pytest.mark.parametrize("key", SCENES)
def test_fixture_vectors_through_the_device_functions(pairs, key):
    sc, r, o = pairs[key]
    sv = VEC["scenes"][key]
    n, dpdu, uv = [0.0, 1.0, 0.0], [1.0, 0.0, 0.0], [0.3, 0.6]
    rows = [[bits(b["mat"])] + b["wo"] + n + dpdu + uv + b["u"] + [0.0] for b in sv["bsdf"]]
    got = r.eval_function("sample_bsdf", rows)
    for b, g in zip(sv["bsdf"], got):
        assert same(g, b["sample"]), (key, b, g)
    rows = [[bits(b["mat"])] + b["wo"] + b["sample"][:3] + n + dpdu + uv + [0.0] for b in sv["bsdf"]]
    got = r.eval_function("eval_bsdf", rows)
    for b, g in zip(sv["bsdf"], got):
        assert same(g, b["eval"]), (key, b, g)
    if sv["lights"]:
        rows = [[bits(l["idx"]), 0.1, 0.5, 0.2, 0.37, 0.58] for l in sv["lights"]]
        got = r.eval_function("sample_light", rows)
        for l, g in zip(sv["lights"], got):
            assert same(g, l["sample"]), (key, l, g)
    rows = [[bits(c["x"]), bits(c["y"]), bits(c["iter"])] for c in sv["camera_rays"]]
    got = r.eval_function("camera_ray", rows)
    for c, g in zip(sv["camera_rays"], got):
        assert same(g, c["ray"]), (key, c, g)


@pytest.mark.parametrize("key", SCENES)
def test_random_inputs_against_the_oracle(pairs, key):
    sc, r, o = pairs[key]
    rng = np.random.default_rng(20 + SCENES.index(key))
    n_mat = sc.info.n_materials
    per = 600
    for mat in range(n_mat):
        wo, nrm, dpdu = unit(rng, per), unit(rng, per), unit(rng, per)   # (dpdu is not made orthogonal to n: a sphere's is not either)
        # half of the directions on the normal's side, grazing ones included
        flip = (np.einsum("ij,ij->i", wo, nrm) < 0) & (rng.uniform(size=per) < 0.5)
        wo[flip] = -wo[flip]
        uv = rng.uniform(-1.5, 2.5, (per, 2)).astype(np.float32)
        u = rng.uniform(size=(per, 3)).astype(np.float32)
        u[::50] = [1.0, 1.0, 1.0]        # the generator can return exactly 1
        u[1::50, 0] = 0.0
        m = np.full((per, 1), bits(mat), np.float32)
        got = r.eval_function("sample_bsdf", np.hstack([m, wo, nrm, dpdu, uv, u, np.zeros((per, 1), np.float32)]))
        want = np.array([o.bsdf_sample(mat, wo[i], nrm[i], dpdu[i], uv[i], u[i]) for i in range(per)])
        assert same(got, want), (key, mat, np.argwhere(~((got == want) | (np.isnan(got) & np.isnan(want))))[:4])
        # evaluation: for the sampled direction (finite ones) and for directions of any kind
        wi = np.where(np.isfinite(want[:, :3]).all(axis=1, keepdims=True), want[:, :3], unit(rng, per))
        wi[::3] = unit(rng, len(wi[::3]))
        got = r.eval_function("eval_bsdf", np.hstack([m, wo, wi, nrm, dpdu, uv, np.zeros((per, 1), np.float32)]))
        want = np.array([o.bsdf_eval(mat, wo[i], wi[i], nrm[i], dpdu[i], uv[i]) for i in range(per)])
        assert same(got, want), (key, mat, np.argwhere(~((got == want) | (np.isnan(got) & np.isnan(want))))[:4])
    for idx in range(min(o.n_lights, 6)):
        pos = rng.uniform(-2, 2, (200, 3)).astype(np.float32)
        u2 = rng.uniform(size=(200, 2)).astype(np.float32)
        got = r.eval_function("sample_light", np.hstack([np.full((200, 1), bits(idx), np.float32), pos, u2]))
        want = np.array([o.light_sample(idx, pos[i], u2[i]) for i in range(200)])
        assert same(got, want), (key, idx)
    xs, ys = rng.integers(0, sc.width, 300), rng.integers(0, sc.height, 300)
    its = rng.integers(1, 5000, 300)
    rows = np.array([[bits(x), bits(y), bits(i)] for x, y, i in zip(xs, ys, its)], np.float32)
    cam = r.eval_function("camera_ray", rows)
    want = np.array([o.camera_ray(int(x), int(y), int(i)) for x, y, i in zip(xs, ys, its)])
    assert same(cam, want), key
    # the intersection record the shade stage rebuilds from (ray, t, primitive), for the hits of those camera rays
    hits, rows = [], []
    for ray in want:
        pid, h = o.trace(ray[:3], ray[3:], sc.info.epsilon)
        if pid >= 0:
            hits.append(h[1:12])
            rows.append(list(ray) + [h[0], bits(pid)])
    assert len(rows) > 50
    rec = r.eval_function("hit_record", rows)
    assert same(rec[:, :11], np.array(hits, np.float32)), key
