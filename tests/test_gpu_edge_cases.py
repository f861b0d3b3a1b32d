"""GPU parity for the camera variants of Camera::GeneratePrimaryRay (Camera.h:45-85) and both
tonemaps of Output (pathtracer.cu:196-213, 2512-2516): thin lens, lat-long environment camera,
gamma instead of filmic.  Same bars as test_gpu_parity: primary ids bit-exact, film equal."""
import json
import os

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu

VARIANTS = {
    "thin_lens": {"apertureRadius": 0.08, "focalDistance": 6.0},
    "environment_camera": {"environment": True},
    "gamma_tonemap": {"filmicTonemap": False},
    "lens_and_gamma": {"apertureRadius": 0.03, "focalDistance": 4.5, "filmicTonemap": False, "fov": 35.0},
}


@pytest.mark.parametrize("key", sorted(VARIANTS))
def test_camera_variant(caster, scenes, key):
    base = scenes["cornell_pt"]
    doc = json.load(open(base))
    doc["camera"].update(VARIANTS[key])
    path = os.path.join(os.path.dirname(base), "cam_%s.json" % key)
    json.dump(doc, open(path, "w"))
    o = ol.Oracle(path)
    sc = caster.Scene(path)
    r = caster.Renderer(sc, 0, 2, 2)
    for it in (1, 5):
        np.testing.assert_array_equal(r.primary_ids(it), o.primary_ids(it))
    spp = 6
    r.render(1, spp, True)
    gpu = r.read_film(2)
    tone = r.read_film(1)
    r.close()
    cpu = o.render(1, spp, 8)
    assert cpu.mean() > 1e-3
    d = (gpu - cpu) / (np.abs(cpu) + 1e-2)
    assert float(np.sqrt(np.mean(d * d))) <= 1e-3
    np.testing.assert_array_equal(gpu, cpu)
    want = o.tonemap(cpu / np.float32(spp), bool(doc["camera"].get("filmicTonemap", True)))
    np.testing.assert_allclose(tone, want, rtol=2e-6, atol=1e-7)


def _write(tmp_path, name, doc, files=()):
    for fn, text in files:
        (tmp_path / fn).write_text(text)
    p = tmp_path / name
    p.write_text(json.dumps(doc))
    return str(p)


QUAD = """v -1 0 -1\nv 1 0 -1\nv 1 0 1\nv -1 0 1\nvn 0 1 0\nf 1//1 3//1 2//1\nf 1//1 4//1 3//1\n"""


def test_scene_without_lights_and_tiny_images(caster, tmp_path):
    """No emitter at all (the light-sampling numbers are still drawn; everything stays black except
    what the camera sees of nothing), a 1x1 image, and an image smaller than one 8x8 tile."""
    for w, h in ((1, 1), (5, 3), (33, 17)):
        doc = {"screen_width": w, "screen_height": h, "integrator": "pt", "maxDepth": 4,
               "camera": {"position": [0, 1.5, 4], "lookat": [0, 0, 0], "up": [0, 1, 0], "fov": 40},
               "material": [{"name": "m", "bsdf": "lambertian", "diffuse": [0.7, 0.6, 0.5]}],
               "scene": [{"mesh": "q.obj", "material": "m"}, {"sphere": True, "center": [0, 0.4, 0], "radius": 0.4, "material": "m"}]}
        path = _write(tmp_path, "nolight_%dx%d.json" % (w, h), doc, [("q.obj", QUAD)])
        o = ol.Oracle(path)
        sc = caster.Scene(path)
        assert sc.info.n_lights == 0
        r = caster.Renderer(sc, 0, 0)
        np.testing.assert_array_equal(r.primary_ids(2), o.primary_ids(2))
        r.render(1, 5, True)
        gpu = r.read_film(2)
        st = r.stats()
        r.close()
        np.testing.assert_array_equal(gpu, o.render(1, 5, 2))
        assert st["samples"] == w * h * 5 and st["stack_overflow"] == 0


def test_empty_row_band_and_zero_iterations(caster, scenes):
    sc = caster.Scene(scenes["cornell_pt"])
    r = caster.Renderer(sc, 0, 2)
    r.set_rows(16, 16)          # an empty band: nothing is traced, the film stays zero
    r.render(1, 3, True)
    assert not r.read_film(2).any() and r.stats()["samples"] == 0
    r.set_rows(0, sc.height)
    r.render(4, 0, False)       # zero iterations: a no-op
    assert r.iterations == 3
    with pytest.raises(caster.CasterError):
        r.set_rows(8, 4)
    r.close()
