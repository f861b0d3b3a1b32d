import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def caster():
    """The product package (its directory name is not a Python identifier)."""
    return importlib.import_module("cuda-pathtracer_amd")


# name -> (generator, kwargs): small versions of the benchmark scenes
SMALL_SCENES = {
    "cornell_ao": dict(name="cornell_ao", width=96, height=96),
    "cornell_pt": dict(name="cornell_pt", width=96, height=96),
    "cornell_vpt": dict(name="cornell_vpt", width=96, height=96, depth=8),
    "cornell_meshlight": dict(name="cornell_meshlight", width=96, height=96),
    "cornell_smoke": dict(name="cornell_smoke", width=96, height=96, depth=8),
    "cornell_lt": dict(name="cornell_lt", width=96, height=96),
    "cornell_fog_lt": dict(name="cornell_fog_lt", width=96, height=96, depth=8),
    "cornell_smoke_lt": dict(name="cornell_smoke_lt", width=96, height=96, depth=8),
    "cornell_bdpt": dict(name="cornell_bdpt", width=96, height=96),
    "cornell_fog_bdpt": dict(name="cornell_fog_bdpt", width=96, height=96),
    "cornell_smoke_bdpt": dict(name="cornell_smoke_bdpt", width=96, height=96),
    # the benchmark scene (all six materials, textures, lines and spheres) under the splatting integrators
    "car_lt": dict(name="car_pt", width=96, height=54, detail=0.01, integrator="lt"),
    "car_bdpt": dict(name="car_pt", width=96, height=54, detail=0.01, integrator="bdpt"),
    "stairs_bdpt": dict(name="stairs_pt", width=64, height=36, detail=0.08, integrator="bdpt"),
    # emitters plus an environment map: the light choice can fall into the environment's bin
    "car_env_lt": dict(name="car_pt", width=96, height=54, detail=0.01, env=True, integrator="lt"),
    "car_env_bdpt": dict(name="car_pt", width=96, height=54, detail=0.01, env=True, integrator="bdpt"),
    # the environment map alone: no emissive triangle to start a light path from
    "car_envonly_lt": dict(name="car_pt", width=96, height=54, detail=0.01, env=True, integrator="lt", drop_emitters=True),
    "car_envonly_bdpt": dict(name="car_pt", width=96, height=54, detail=0.01, env=True, integrator="bdpt", drop_emitters=True),
    "knot_pt": dict(name="knot_pt", width=96, height=96, detail=0.08),
    "stairs_pt": dict(name="stairs_pt", width=128, height=72, detail=0.08),
    "car_pt": dict(name="car_pt", width=128, height=72, detail=0.01),
    "car_env": dict(name="car_pt", width=128, height=72, detail=0.01, env=True),
    "odd_size": dict(name="cornell_pt", width=70, height=45),
}


@pytest.fixture(scope="session")
def scenes(tmp_path_factory):
    import gen_scenes
    base = tmp_path_factory.mktemp("scenes")
    out = {}
    for key, kw in SMALL_SCENES.items():
        kw = dict(kw)
        name = kw.pop("name")
        integrator = kw.pop("integrator", None)
        drop_emitters = kw.pop("drop_emitters", False)
        out[key] = gen_scenes.generate(str(base / key), name, **kw)
        if integrator:   # the same scene under another of the reference's integrators
            import json
            doc = json.load(open(out[key]))
            doc["integrator"] = integrator
            if drop_emitters:
                doc["light"] = [l for l in doc["light"] if "infinite" in l]
            json.dump(doc, open(out[key], "w"))
    return out
