"""GPU parity: the HIP renderer, called through the C-ABI, against the CPU oracle
on the same seeded samples.

Bars: primary-hit primitive ids bit-exact; the accumulated film equal to the
oracle's.  Because both sides draw the same per-pixel random streams and share
the floating-point contract of fpspec.h, the film is expected to agree far inside
the stated tolerance: per-pixel relative RMSE (|gpu-cpu| / (cpu + 1e-2)) <= 1e-3,
and at most 0.2 % of pixels may differ at all by more than 1e-5 relative.
"""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu

SPP = {"cornell_ao": 8, "cornell_pt": 8, "cornell_vpt": 4, "cornell_smoke": 4, "cornell_meshlight": 6, "knot_pt": 4, "stairs_pt": 4, "car_pt": 4, "car_env": 4,
       "odd_size": 4}


def rel_rmse(gpu, cpu):
    d = (gpu - cpu) / (np.abs(cpu) + 1e-2)
    return float(np.sqrt(np.mean(d * d)))


@pytest.mark.parametrize("key", sorted(SPP))
def test_primary_ids_bit_exact(caster, scenes, key):
    o = ol.Oracle(scenes[key])
    sc = caster.Scene(scenes[key])
    r = caster.Renderer(sc, 0, 2)
    for it in (1, 7):
        cpu = o.primary_ids(it)
        gpu = r.primary_ids(it)
        assert cpu.shape == gpu.shape
        assert np.array_equal(cpu, gpu), "%d of %d ids differ" % ((cpu != gpu).sum(), cpu.size)
    r.close()


@pytest.mark.parametrize("key", sorted(SPP))
def test_film_matches_oracle(caster, scenes, key):
    spp = SPP[key]
    o = ol.Oracle(scenes[key])
    cpu = o.render(1, spp, 8)
    sc = caster.Scene(scenes[key])
    r = caster.Renderer(sc, 0, 3)   # 3 iterations in flight: exercises batching (spp is not a multiple)
    r.render(1, spp, True)
    gpu = r.read_film(2)
    st = r.stats()
    r.close()
    assert st["stack_overflow"] == 0
    assert st["rays"][0] >= sc.width * sc.height * spp and st["samples"] == sc.width * sc.height * spp
    assert np.isfinite(gpu).all()
    assert cpu.mean() > 1e-3 and gpu.mean() > 1e-3   # not a comparison of two empty films
    differ = np.abs(gpu - cpu) > 1e-5 * (np.abs(cpu) + 1e-2)
    frac = differ.any(axis=2).mean()
    err = rel_rmse(gpu, cpu)
    print(key, "relRMSE %.3g, pixels differing %.4f%%, exact-equal %s" % (err, 100 * frac, np.array_equal(gpu, cpu)))
    assert err <= 1e-3
    assert frac <= 0.002


def test_traversal_kernel_against_oracle_ray_by_ray(caster, scenes):
    """caster_trace_rays (the renderer's traversal kernel on rays of the caller's) returns the oracle's
    closest hit -- same primitive, same distance to the bit -- for rays from all over the staircase,
    bounded and unbounded, of unit and other lengths, and for a tmax that is not a number (which bounds
    nothing: every comparison with it fails)."""
    import ctypes as C
    sc = caster.Scene(scenes["stairs_pt"])
    o = ol.Oracle(scenes["stairs_pt"])
    rng = np.random.RandomState(9)
    n = 6000
    org = (rng.rand(n, 3) * [6.0, 4.0, 6.0] + [-3.0, 0.2, -3.0]).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[::7] *= 1.5                                                    # not every direction the integrators make is a unit vector
    d = d.astype(np.float32)
    tmax = np.full(n, np.inf, np.float32)
    tmax[1::3] = (rng.rand(len(tmax[1::3])) * 4).astype(np.float32)
    tmax[5::50] = np.nan
    r = caster.Renderer(sc, 0, 1)
    prim, t = r.trace_rays(org, d, tmax)
    r.close()
    o.L.oracle_trace.restype = C.c_int
    hits = 0
    for i in range(n):
        out = (C.c_float * 12)()
        p = o.L.oracle_trace(o.h, (C.c_float * 3)(*org[i]), (C.c_float * 3)(*d[i]), C.c_float(sc.info.epsilon), C.c_float(tmax[i]), out)
        assert p == prim[i], (i, p, prim[i], tmax[i])
        if p >= 0:
            hits += 1
            assert out[0] == t[i], (i, out[0], t[i])
    assert n // 3 < hits < n
    assert (prim[5::50] >= 0).sum() > 10                             # the unbounded-by-NaN rays hit things


def test_traversal_kernel_rays_parallel_to_an_axis(caster, scenes):
    """Rays with one or two direction components of exactly zero (1 / direction is then infinite), many of them
    starting in a plane of a node's box, where (plane - origin) / 0 is not a number and the slab test's minima and
    maxima decide by which operand they drop: the traversal kernel keeps such rays off its octant-addressed slab
    test, and every hit equals the oracle's."""
    import ctypes as C
    sc = caster.Scene(scenes["stairs_pt"])
    o = ol.Oracle(scenes["stairs_pt"])
    boxes, meta, _ = sc.bvh()
    rng = np.random.RandomState(21)
    n = 4000
    pick = boxes[rng.randint(0, len(boxes), n)]
    org = (rng.rand(n, 3) * [6.0, 4.0, 6.0] + [-3.0, 0.2, -3.0]).astype(np.float32)
    d = rng.normal(size=(n, 3))
    axis = rng.randint(0, 3, n)
    d[np.arange(n), axis] = 0.0                                      # parallel to the planes of that axis
    two = rng.rand(n) < 0.3
    d[two, (axis[two] + 1) % 3] = 0.0                                 # along a coordinate axis
    d[rng.rand(n) < 0.25] *= -1.0
    d[::5][d[::5] == 0.0] = -0.0                                      # both signs of zero
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(np.float32)
    on_plane = rng.rand(n) < 0.7
    side = rng.randint(0, 2, n)
    rows = np.arange(n)[on_plane]
    org[rows, axis[rows]] = pick[rows, axis[rows] + 3 * side[rows]]   # the origin lies in a lo or hi plane of some box
    inside = rng.rand(n) < 0.5                                        # and for half of them within that box otherwise
    for a in range(3):
        m = on_plane & inside & (axis != a)
        org[m, a] = (pick[m, a] + (pick[m, a + 3] - pick[m, a]) * rng.rand(m.sum())).astype(np.float32)
    tmax = np.full(n, np.inf, np.float32)
    tmax[1::4] = (rng.rand(len(tmax[1::4])) * 4).astype(np.float32)
    r = caster.Renderer(sc, 0, 1)
    prim, t = r.trace_rays(org, d, tmax)
    r.close()
    o.L.oracle_trace.restype = C.c_int
    hits = 0
    for i in range(n):
        out = (C.c_float * 12)()
        p = o.L.oracle_trace(o.h, (C.c_float * 3)(*org[i]), (C.c_float * 3)(*d[i]), C.c_float(sc.info.epsilon), C.c_float(tmax[i]), out)
        assert p == prim[i], (i, p, prim[i], org[i], d[i])
        if p >= 0:
            hits += 1
            assert out[0] == t[i], (i, out[0], t[i])
    assert n // 4 < hits < n
