"""The HIP renderer, through the C-ABI, against answers that come from neither the reference's text nor the CPU
oracle: the form-factor integral of a rectangular emitter over a Lambertian floor, and the closed-form radiance of a
convex body inside a uniform environment (scenes and formulas: test_oracle_independent.py).  The other GPU tests
compare with the oracle; these hold the product itself against textbook radiometry."""
import numpy as np
import pytest

import test_oracle_independent as ind

pytestmark = pytest.mark.gpu


def render_mean(caster, path, spp):
    sc = caster.Scene(path)
    r = caster.Renderer(sc, 0, 8)
    r.render(1, spp, True)
    img = r.read_film(0)          # linear mean, row 0 = bottom scanline
    st = r.stats()
    r.close()
    assert st["stack_overflow"] == 0 and st["invalid"] == 0
    return img


def test_direct_light_of_a_rectangle_matches_the_form_factor(caster, tmp_path):
    path, patch, want = ind.form_factor_scene(str(tmp_path))
    img = render_mean(caster, path, 1024)
    got = float(img[patch].mean())
    print("floor radiance", got, "form factor", want)
    assert abs(got - want) <= 0.005 * want, (got, want)


@pytest.mark.parametrize("bsdf,want", ind.FURNACE)
def test_a_convex_body_in_a_uniform_environment_shows_its_albedo(caster, tmp_path, bsdf, want):
    path, body, corner = ind.furnace_scene(str(tmp_path), bsdf)
    img = render_mean(caster, path, 512)
    assert abs(float(img[corner].mean()) - 1.0) < 1e-5
    got = float(img[body].mean())
    print(bsdf, "sphere in the furnace", got, "expected", want)
    assert abs(got - want) <= 0.005 * want, (bsdf, got, want)


@pytest.mark.parametrize("boundary", ["matched", "glass", "grid"])
def test_a_scattering_medium_without_absorption_in_a_uniform_environment_shows_one(caster, tmp_path, boundary):
    path, body = ind.volume_furnace_scene(str(tmp_path), boundary)
    sc = caster.Scene(path)
    r = caster.Renderer(sc, 0, 8)
    r.render(1, 1024, True)
    img = r.read_film(0)
    st = r.stats()
    r.close()
    assert st["stack_overflow"] == 0
    got = img[body].reshape(-1, 3).mean(axis=0)
    print(boundary, "cube of fog in the furnace", got, "invalid samples", st["invalid"])
    assert np.all(np.abs(got - 1.0) <= 0.03), got


def test_an_emitter_seen_through_an_absorbing_medium_is_attenuated_twice_as_in_the_reference(caster, tmp_path):
    """(pathtracer.cu:1093-1101 applies the transmittance on top of a throughput that already carries it: see the
    oracle's test of the same name.)"""
    path, patch, want = ind.absorption_scene(str(tmp_path))
    img = render_mean(caster, path, 1024)
    got = img[patch].reshape(-1, 3).mean(axis=0)
    twice = want * want / 3.0
    print("through the medium", got, "attenuated twice", twice)
    assert np.all(np.abs(got - twice) <= 0.02 * twice), (got, twice)


@pytest.mark.parametrize("wall,want", [(False, np.pi), (True, np.pi / 2)])
def test_ambient_occlusion_of_an_open_floor_and_at_the_foot_of_a_wall(caster, tmp_path, wall, want):
    img = render_mean(caster, ind.ao_scene(str(tmp_path), wall), 1024)
    got = float(img.mean())
    print("AO", "at the foot of a wall" if wall else "open floor", got, "expected", want)
    # (rays through the diagonal the two floor triangles share may miss both: see the oracle's version of this test)
    assert abs(got - want) <= (0.01 if wall else 5e-4) * want


def test_path_tracing_light_tracing_and_bidirectional_path_tracing_estimate_the_same_image(caster, tmp_path):
    """The three integrators of the HIP renderer -- queues (pt), splatting through queues or in place (lt), one
    kernel per sample with vertex columns in HBM (bdpt) -- on the scene of the oracle's test of the same name."""
    films = {}
    for integ, spp in (("pt", 1024), ("lt", 4096), ("bdpt", 512)):
        sub = tmp_path / integ
        sub.mkdir()
        films[integ] = render_mean(caster, ind.lit_floor_scene(str(sub), integ), spp)
    pt = ind.block_means(films["pt"])
    lit = pt.sum(axis=2) > 0.1 * pt.sum(axis=2).max()
    assert lit.sum() >= 8
    for other in ("lt", "bdpt"):
        ot = ind.block_means(films[other])
        ratio = ot[lit].sum() / pt[lit].sum()
        rel = np.abs(ot - pt).sum(axis=2)[lit] / pt.sum(axis=2)[lit]
        print(other, "against pt: total", ratio, "median block difference", np.median(rel), "worst", rel.max())
        assert 0.98 < ratio < 1.02, (other, ratio)
        assert np.median(rel) < 0.06 and rel.max() < 0.25, (other, rel)   # (block-wise the difference is noise)


@pytest.mark.parametrize("how", ["translate-rotate-scale", "transform"])
def test_a_mesh_lands_where_its_transformation_puts_it(caster, tmp_path, how):
    """The traversal kernel's closest hits in a scene whose mesh is placed by a transformation, against the exhaustive
    float64 search over the triangles transformed with textbook matrices (the oracle's test of the same name)."""
    path, world, world_n, centre = ind.transformed_mesh_scene(str(tmp_path), how)
    sc = caster.Scene(path)
    r = caster.Renderer(sc, 0, 8)      # (the hook traces as many rays as the renderer has path slots: 8 x 16 x 16)
    org, dr = ind.rays_about(centre, 1500)
    prim, t = r.trace_rays(org, dr, np.full(len(org), np.inf, np.float32))
    rec = r.eval_function("hit_record", np.concatenate([org, dr, t[:, None], prim.view(np.float32)[:, None]], 1))
    r.close()
    wrong, hits = 0, 0
    for i in range(len(org)):
        want, want_t = ind.closest_hit_f64(org[i].astype(np.float64), dr[i].astype(np.float64), world, [], 1e-3, np.inf)
        if (prim[i] >= 0) != (want >= 0) or (want >= 0 and abs(t[i] - want_t) > 3e-4 * (1 + want_t)):
            wrong += 1
            continue
        if want >= 0:
            hits += 1
            assert abs(abs(float(rec[i, 3:6].astype(np.float64) @ world_n[want])) - 1.0) < 1e-4
    assert hits > len(org) // 15
    assert wrong <= len(org) * 0.006, "%d of %d rays disagree with the float64 search" % (wrong, len(org))


def test_each_of_three_emitters_lights_the_floor_by_its_own_form_factor(caster, tmp_path):
    """(The oracle's test_lights_are_picked_in_proportion_to_their_power, without the count of picks: the film of the
    one-pixel image after 262 144 iterations, channel by channel, against each emitter's form-factor integral.)"""
    path, quads, share = ind.three_emitters_scene(str(tmp_path))
    sc = caster.Scene(path)
    r = caster.Renderer(sc, 0, 4096)
    n = 1 << 18
    r.render(1, n, True)
    got = r.read_film(0).reshape(3)
    cam = r.eval_function("camera_ray", np.array([[0, 0, it] for it in range(1, 257)], np.int32).view(np.float32))
    r.close()
    want, _ = ind.three_emitters_expected(quads, ind.floor_landing(cam))
    print("direct light per emitter", got, "form factors", want)
    assert np.allclose(got, want, rtol=0.02), (got, want)


def test_a_textured_floor_in_a_uniform_environment_shows_its_texture(caster, tmp_path):
    path, size, albedo = ind.textured_floor_scene(str(tmp_path))
    img = render_mean(caster, path, 1024)
    sc = caster.Scene(path)
    r = caster.Renderer(sc, 0, 1)
    W, H = size
    xy = np.array([[x, y, 1] for y in range(H) for x in range(W)], np.int32)
    cam = r.eval_function("camera_ray", xy.view(np.float32)).reshape(H, W, 6)
    r.close()
    checked, worst = ind.textured_floor_worst_block(img, size, albedo, lambda x, y: cam[y, x])
    print("blocks", checked, "worst relative difference", worst)
    assert checked >= 12
    assert worst < 0.03
