"""The entry point that allocates nothing (caster_trace_rays_static) against caster_trace_rays.

Both trace the caller's rays with the renderer's traversal kernel; the static one takes device
arrays and keeps its ray queue and work counters in a static device array instead of the
renderer's allocated queues.  Same primitive ids and distances, bit for bit, on two calls in a row
(the second smaller, so it runs over the first call's leftovers in the shared array)."""
import numpy as np
import pytest

import gen_scenes

pytestmark = pytest.mark.gpu


def _rays(rng, n, scene_lo, scene_hi):
    o = rng.uniform(scene_lo, scene_hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    tmax = np.where(rng.uniform(size=n) < 0.7, np.inf, rng.uniform(0.05, 3.0, n)).astype(np.float32)
    return o, d.astype(np.float32), tmax


def test_static_entry_point_matches_trace_rays(caster, tmp_path):
    import torch
    path = gen_scenes.generate(str(tmp_path / "car"), "car_pt", 0.05, 256, 144, 8)
    sc = caster.Scene(path)
    r = caster.Renderer(sc, 0, 8, 1)
    rng = np.random.default_rng(7)
    dev = torch.device("cuda", 0)
    for n in (200_000, 3_001):   # the second call is smaller than the first
        o, d, tmax = _rays(rng, n, -3.0, 3.0)
        want_prim, want_t = r.trace_rays(o, d, tmax)
        assert (want_prim >= 0).sum() > n // 10
        to, td, tt = (torch.from_numpy(a).to(dev) for a in (o, d, tmax))
        prim = torch.full((n + 16,), -7, dtype=torch.int32, device=dev)
        t = torch.full((n + 16,), -7.0, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        r.trace_rays_static(n, to.data_ptr(), td.data_ptr(), tt.data_ptr(), prim.data_ptr(), t.data_ptr())
        got_prim, got_t = prim.cpu().numpy(), t.cpu().numpy()
        np.testing.assert_array_equal(got_prim[:n], want_prim)
        np.testing.assert_array_equal(got_t[:n].view(np.uint32), want_t.view(np.uint32))
        assert (got_prim[n:] == -7).all() and (got_t[n:] == -7.0).all()   # nothing written past n
    assert r.stats()["stack_overflow"] == 0
    # more rays than the static array holds: a clear error, nothing traced
    with pytest.raises(caster.CasterError) as e:
        r.trace_rays_static(caster.STATIC_MAX_RAYS + 1, to.data_ptr(), td.data_ptr(), tt.data_ptr(), prim.data_ptr(),
                            t.data_ptr())
    assert e.value.code == 1 and "CASTER_STATIC_MAX_RAYS" in str(e.value)
    r.close()
