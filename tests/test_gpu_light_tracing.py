"""GPU parity of the light-tracing integrator (pathtracer.cu:1232-1375) against the CPU oracle.

Every light path draws the same random stream on both sides and every splat value is computed
under the same floating-point contract, so the two films are sums of the SAME numbers.  What
differs is the order of the additions: the GPU adds splats with float atomics as they are produced
(the reference does too, pathtracer.cu:1306-1308), the oracle adds them path by path.  The bar
therefore is: first-hit primitive ids of the light paths bit-exact; per-pixel relative RMSE
(|gpu-cpu| / (|cpu| + 1e-2)) <= 1e-5; at most 0.2 % of the pixels differing by more than 1e-4 relative.
"""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu

SPP = {"cornell_lt": 8, "cornell_fog_lt": 6, "cornell_smoke_lt": 6, "car_lt": 6, "car_env_lt": 4, "car_envonly_lt": 3}


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
def test_first_hits_of_light_paths_bit_exact(caster, scenes, key):
    o = ol.Oracle(scenes[key])
    sc = caster.Scene(scenes[key])
    assert caster.INTEGRATOR_NAMES[sc.integrator] == "lt"
    r = caster.Renderer(sc, 0, 2)
    for it in (1, 5):
        cpu = o.primary_ids(it)
        gpu = r.primary_ids(it)
        if "envonly" in key:
            assert (cpu < 0).all()                                  # no light path starts
        else:
            assert (cpu >= 0).mean() > (0.5 if key.startswith("cornell") else 0.05)   # most light paths reach the box (its front is open)
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
    # light tracing leaves most of the film to indirect splats: the comparison is not one of a few bright pixels
    lit = float((cpu.max(axis=2) > 0).mean())   # (mirror and glass get no splats: Fr is zero for them)
    if "envonly" in key:
        assert lit == 0.0 and gpu.max() == 0.0                      # nothing to splat
    else:
        assert lit > (0.75 if key.startswith("cornell") else 0.05), lit
    check_close(gpu, cpu, key)
    # the transmittance walks of the camera connections, segment by segment
    assert st["rays"][3] == counts.as_dict()["rays"][3]


def test_streams_batches_and_row_bands_add_up(caster, scenes):
    key, spp = "cornell_fog_lt", 6
    sc = caster.Scene(scenes[key])
    ref = None
    for iters_in_flight, streams in ((6, 1), (2, 3), (1, 2)):
        r = caster.Renderer(sc, 0, iters_in_flight, streams)
        r.render(1, spp, True)
        film = r.read_film(2)
        r.close()
        if ref is None:
            ref = film
        else:
            check_close(film, ref, "%d in flight x %d streams" % (iters_in_flight, streams))
    # a row band selects which light paths are traced, not where they land: the bands' films add up
    total = np.zeros_like(ref)
    for y0, y1 in ((0, 40), (40, 96)):
        r = caster.Renderer(sc, 0, 3)
        r.set_rows(y0, y1)
        r.render(1, spp, True)
        total += r.read_film(2)
        r.close()
    check_close(total, ref, "two row bands")
