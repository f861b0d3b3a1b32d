"""GPU parity of the bidirectional integrator (pathtracer.cu:1379-1956) against the CPU oracle.

Every sample draws the same random stream on both sides (camera walk, light walk, then the joins
in (s, t) order) and every contribution is computed under the same floating-point contract, so
the two films are sums of the SAME numbers.  What differs is the order of the additions: the GPU
adds with float atomics (the reference does too, pathtracer.cu:1944-1953) and gives a sample's
own pixel the sum of its joins in one add, the oracle adds join by join, thread buffer by thread
buffer.  The bar therefore is: primary hit ids bit-exact; ray counts (walk rays and transmittance
segments) equal; per-pixel relative RMSE (|gpu-cpu| / (|cpu| + 1e-2)) <= 1e-5; at most 0.2 % of
the pixels differing by more than 1e-4 relative.
"""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu

SPP = {"cornell_bdpt": 6, "cornell_fog_bdpt": 4, "cornell_smoke_bdpt": 4, "car_bdpt": 4, "stairs_bdpt": 3, "car_env_bdpt": 3,
       "car_envonly_bdpt": 3}


def rel(gpu, cpu):
    return (gpu - cpu) / (np.abs(cpu) + 1e-2)


def check_close(gpu, cpu, what):
    d = rel(gpu, cpu)
    err = float(np.sqrt(np.mean(d * d)))
    frac = (np.abs(d) > 1e-4).any(axis=2).mean()
    print(what, "relRMSE %.3g, pixels beyond 1e-4: %.4f%%, max %.3g" % (err, 100 * frac, np.abs(d).max()))
    assert err <= 1e-5
    assert frac <= 0.002


@pytest.mark.parametrize("key", sorted(SPP))
def test_primary_ids_bit_exact(caster, scenes, key):
    o = ol.Oracle(scenes[key])
    sc = caster.Scene(scenes[key])
    assert caster.INTEGRATOR_NAMES[sc.integrator] == "bdpt"
    r = caster.Renderer(sc, 0, 2)
    for it in (1, 3):
        cpu = o.primary_ids(it)
        gpu = r.primary_ids(it)
        assert np.array_equal(cpu, gpu), "%d of %d ids differ" % ((cpu != gpu).sum(), cpu.size)
    r.close()


@pytest.mark.parametrize("key", sorted(SPP))
def test_film_matches_oracle(caster, scenes, key):
    spp = SPP[key]
    o = ol.Oracle(scenes[key])
    counts = ol.Counts()
    cpu = o.render(1, spp, 8, counts=counts)
    sc = caster.Scene(scenes[key])
    r = caster.Renderer(sc, 0, 4)   # batches of 4 iterations: spp is not a multiple
    r.render(1, spp, True)
    gpu = r.read_film(2)
    st = r.stats()
    r.close()
    assert st["stack_overflow"] == 0 and st["samples"] == sc.width * sc.height * spp
    assert np.isfinite(gpu).all() and np.isfinite(cpu).all()
    if "envonly" not in key:   # no emissive triangle: whatever the reference's conventions leave, on both sides
        assert cpu.mean() > 1e-3 and gpu.mean() > 1e-3
    check_close(gpu, cpu, key)
    # the same walks on both sides: sub-path rays and transmittance segments, one by one
    c = counts.as_dict()
    assert st["rays"][0] == c["rays"][0]
    assert st["rays"][3] == c["rays"][3]


def test_streams_batches_and_row_bands_add_up(caster, scenes):
    key, spp = "cornell_fog_bdpt", 4
    sc = caster.Scene(scenes[key])
    ref = None
    for iters_in_flight, streams in ((4, 1), (2, 3), (1, 2)):
        r = caster.Renderer(sc, 0, iters_in_flight, streams)
        r.render(1, spp, True)
        film = r.read_film(2)
        r.close()
        if ref is None:
            ref = film
        else:
            check_close(film, ref, "%d in flight x %d streams" % (iters_in_flight, streams))
    # a row band selects which samples are traced; their pinhole joins land anywhere: the bands' films add up
    total = np.zeros_like(ref)
    for y0, y1 in ((0, 40), (40, 96)):
        r = caster.Renderer(sc, 0, 2)
        r.set_rows(y0, y1)
        r.render(1, spp, True)
        total += r.read_film(2)
        r.close()
    check_close(total, ref, "two row bands")


@pytest.mark.parametrize("key", ["car_bdpt", "stairs_bdpt"])
def test_joins_on_queues_in_place_and_left_over(caster, scenes, key, monkeypatch):
    """Without a density grid the shadow rays of the joins go through the traversal kernel; rounds of
    samples that find the join queue too full are evaluated afterwards by the kernel that walks them in place.  The
    three ways -- queued (the default), nearly every round left over (one join slot per sample), everything in place
    (CASTER_BDPT_INKERNEL=1) -- give the oracle's film to the integrator's tolerance and the oracle's ray counts."""
    spp = SPP[key]
    o = ol.Oracle(scenes[key])
    counts = ol.Counts()
    cpu = o.render(1, spp, 8, counts=counts)
    c = counts.as_dict()
    sc = caster.Scene(scenes[key])
    for name, env in (("queued", {}), ("left over", {"CASTER_BDPT_JOIN_SLOTS": "1"}), ("in place", {"CASTER_BDPT_INKERNEL": "1"})):
        for k in ("CASTER_BDPT_JOIN_SLOTS", "CASTER_BDPT_INKERNEL"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        r = caster.Renderer(sc, 0, 2, 2)
        r.render(1, spp, True)
        gpu = r.read_film(2)
        st = r.stats()
        kt = r.kernel_times()
        r.close()
        assert st["stack_overflow"] == 0 and st["samples"] == sc.width * sc.height * spp
        check_close(gpu, cpu, "%s, %s" % (key, name))
        assert st["rays"][0] == c["rays"][0], name
        assert st["rays"][3] == c["rays"][3], name
