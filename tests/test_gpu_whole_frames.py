"""Whole frames of the benchmark configurations, GPU against the CPU oracle, every pixel equal.

The other full-size tests compare the first-hit map of the whole frame and the film of a band of
scanlines; these compare the film of the WHOLE frame at a few samples per pixel -- tens of millions of
samples each, through every material, both light-sampling strategies, lines, spheres and the medium --
and the ray statistics with it.  The bar is bit equality: the closest hit is a function of the ray
alone on both sides (smallest distance, then highest primitive index; boxes culled with a slack; a
sphere candidate in front of its own bounding box rejected), so the order in which a wavefront happens
to walk the tree cannot show."""
import importlib
import time

import numpy as np
import pytest

import gen_scenes
import oracle_lib as ol

pytestmark = pytest.mark.gpu

# key -> (generator scene, detail, width, height, maxDepth, samples per pixel)
FRAMES = {
    "C1_cornell_ao_512": ("cornell_ao", 1.0, 512, 512, 8, 16),      # 4.2e6 samples: the whole of configuration 1
    "C2_knot_1024": ("knot_pt", 1.0, 1024, 1024, 8, 8),             # 8.4e6 samples; glass, mirror, checker texture
    "C3_stairs_1080p": ("stairs_pt", 1.0, 1920, 1080, 8, 16),     # 3.3e7 samples; lines and spheres on the railing
    "C4_cornell_fog_1024": ("cornell_vpt", 1.0, 1024, 1024, 16, 8),  # 8.4e6 samples; spheres inside the medium
    "C5_car_1080p": ("car_pt", 1.0, 1920, 1080, 8, 4),             # 8.3e6 samples; 1.37 M primitives
    "X_cornell_smoke_512": ("cornell_smoke", 1.0, 512, 512, 16, 8),  # 2.1e6 samples; a density grid: the vertex in three phases
}


@pytest.mark.parametrize("key", sorted(FRAMES))
def test_whole_frame_equals_oracle(caster, tmp_path, key):
    name, detail, w, h, depth, spp = FRAMES[key]
    path = gen_scenes.generate(str(tmp_path / key), name, detail, w, h, depth)
    sc = caster.Scene(path)
    r = caster.Renderer(sc, 0, 0)
    r.render(1, spp, True)
    gpu = r.read_film(2)
    st = r.stats()
    r.close()
    assert st["stack_overflow"] == 0 and st["samples"] == w * h * spp
    o = ol.Oracle(path)
    counts = ol.Counts()
    cpu = np.zeros_like(gpu)
    t0 = time.time()
    band = 120
    for y0 in range(0, h, band):   # in bands, so that a long run keeps printing
        y1 = min(y0 + band, h)
        cpu[y0:y1] = o.render(1, spp, 16, rows=(y0, y1), counts=counts)[y0:y1]
        print("oracle rows %d of %d after %.0f s" % (y1, h, time.time() - t0), flush=True)
    bad = np.argwhere((gpu != cpu).any(axis=2))
    assert len(bad) == 0, "%d pixels differ, first (y, x): %s" % (len(bad), bad[:8].tolist())
    c = counts.as_dict()
    print("rays gpu %s oracle %s" % (st["rays"], c["rays"]))
    assert st["invalid"] == c["invalid"]
    if sc.integrator == caster.IT_PT:
        # one extension ray per path vertex on both sides (the GPU sends fewer shadow and probe rays:
        # it skips those whose term is zero whatever they hit)
        assert st["rays"][0] == c["rays"][0]


def test_density_grid_vertex_in_phases_equals_the_vertex_in_one_kernel(caster, tmp_path, monkeypatch):
    """A scene with a density grid is rendered with the volumetric vertex cut into three phases around its two
    traced rays (the default), and -- CASTER_VPT_INKERNEL=1, a knob -- with the whole vertex and its walks in one
    kernel: the same film bit for bit, the same extension rays and transmittance segments; the phased vertex sends
    fewer probe rays (those that cannot reach an emitter)."""
    path = gen_scenes.generate(str(tmp_path / "smoke"), "cornell_smoke", 1.0, 256, 256, 16)
    sc = caster.Scene(path)
    out = {}
    for knob in ("0", "1"):
        monkeypatch.setenv("CASTER_VPT_INKERNEL", knob)
        r = caster.Renderer(sc, 0, 4, 2)
        r.render(1, 12, True)
        out[knob] = (r.read_film(2), r.stats())
        r.close()
    monkeypatch.delenv("CASTER_VPT_INKERNEL")
    np.testing.assert_array_equal(out["0"][0], out["1"][0])
    a, b = out["0"][1], out["1"][1]
    assert a["rays"][0] == b["rays"][0] and a["rays"][3] == b["rays"][3] and a["invalid"] == b["invalid"]
    assert 0 < a["rays"][2] < b["rays"][2]


def test_density_grid_in_stripes_bands_and_uneven_batches(caster, tmp_path):
    """The phased density-grid vertex where part of the slots of a batch lie outside the traced rows (interleaved
    stripes of 8 scanlines, a band), where the image is not a multiple of the 8 x 8 tiles, and with batches of 1, 3 and
    5 iterations on 1 to 3 streams: the stripes' films add up to the whole frame's bit for bit, and the whole frame does
    not depend on the batching."""
    sharding = importlib.import_module("cuda-pathtracer_amd.sharding")
    path = gen_scenes.generate(str(tmp_path / "smoke"), "cornell_smoke", 1.0, 203, 149, 16)
    sc = caster.Scene(path)
    spp = 7
    ref = None
    for iters, streams in ((5, 1), (3, 2), (1, 3)):
        r = caster.Renderer(sc, 0, iters, streams)
        r.render(1, spp, True)
        film = r.read_film(2)
        st = r.stats()
        r.close()
        assert st["stack_overflow"] == 0 and st["samples"] == sc.width * sc.height * spp
        if ref is None:
            ref = film
        else:
            np.testing.assert_array_equal(film, ref)
    total = np.zeros_like(ref)
    for rank in range(3):
        r = caster.Renderer(sc, 0, 2, 2)
        r.set_row_stripes(*sharding.row_stripes(sc.height, rank, 3))
        r.render(1, spp, True)
        part = r.read_film(2)
        r.close()
        assert np.count_nonzero(part[total != 0]) == 0   # (disjoint scanlines)
        total += part
    np.testing.assert_array_equal(total, ref)
    r = caster.Renderer(sc, 0, 4, 1)
    r.set_rows(40, 97)
    r.render(1, spp, True)
    band = r.read_film(2)
    r.close()
    np.testing.assert_array_equal(band[40:97], ref[40:97])
    assert not band[:40].any() and not band[97:].any()
