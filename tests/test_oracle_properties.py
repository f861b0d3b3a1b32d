"""Properties of the oracle's scattering functions that do NOT come from transcribing the reference.

The oracle restates the reference's SampleBSDF / Fr line by line, and the HIP lobes are checked against the oracle
bit for bit -- a misreading made once would be made twice.  These tests hold the oracle against what any BSDF
sampler must satisfy: a density integrates to one, the sampled direction's density and value are the ones the
evaluation reports, a reflection BRDF is reciprocal, no lobe creates energy, and the directions drawn follow the
density reported.  Where the reference's own formulas break a property, the case is an expected failure that names
the quirk (pathtracer.cu line numbers):
  * GGX_D uses (1 - cos phi) where sin^2 phi is meant (:80), so the "density" D |wh.n| / (4 wo.wh) of every
    microfacet lobe is not the density its sampler (SampleGGX, :127-147) draws from, and does not integrate to one;
  * Fr's rough dielectric derives the half vector of a REFLECTION from the refraction formula -(ei wo + et wi)
    (:777-816), so evaluating a sampled reflection does not return what the sampler returned.
"""
import json

import numpy as np
import pytest

import oracle_lib as ol

MATS = [
    {"name": "lambert", "bsdf": "lambertian", "diffuse": [0.8, 0.6, 0.4]},
    {"name": "mirror", "bsdf": "mirror", "specular": [0.9, 0.9, 0.9]},
    {"name": "glass", "bsdf": "dielectric", "insideIOR": 1.5, "outsideIOR": 1.0, "specular": [1, 1, 1]},
    {"name": "frost", "bsdf": "roughdielectric", "alpha": 0.3, "insideIOR": 1.5, "outsideIOR": 1.0},
    {"name": "metal", "bsdf": "roughconduct", "alpha": 0.3, "eta": [0.2, 0.92, 1.1], "k": [3.9, 2.45, 2.14]},
    {"name": "floor", "bsdf": "substrate", "alpha": 0.3, "diffuse": [0.4, 0.3, 0.2], "specular": [0.04, 0.04, 0.04]},
    {"name": "brushed", "bsdf": "roughconduct", "alphaU": 0.2, "alphaV": 0.4, "eta": [0.2, 0.92, 1.1], "k": [3.9, 2.45, 2.14]},
]
LAMBERT, MIRROR, GLASS, FROST, METAL, FLOOR, BRUSHED = range(7)
N, DPDU, UV = np.float32([0, 0, 1]), np.float32([1, 0, 0]), np.float32([0.5, 0.5])
QUIRK_D = "GGX_D keeps (1 - cos phi) for sin^2 phi (pathtracer.cu:80): the reported density is not the sampler's"
QUIRK_RD = "Fr takes a reflection's half vector from the refraction formula (pathtracer.cu:777-816)"


@pytest.fixture(scope="module")
def oracle(tmp_path_factory):
    d = tmp_path_factory.mktemp("props")
    doc = {"screen_width": 8, "screen_height": 8, "integrator": "pt", "material": MATS,
           "camera": {"position": [0, 0, 3], "lookat": [0, 0, 0], "up": [0, 1, 0]},
           "scene": [{"sphere": True, "center": [0, 0, 0], "radius": 1, "material": "lambert"}], "light": []}
    (d / "scene.json").write_text(json.dumps(doc))
    return ol.Oracle(str(d / "scene.json"))


def direction(theta, phi):
    return np.float32([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)])


WO = direction(0.6, 0.4)


def sphere_grid(n_theta, n_phi):
    """Midpoint rule over the sphere: directions (n, 3) and solid-angle weights (n,)."""
    th = (np.arange(n_theta) + 0.5) * np.pi / n_theta
    ph = (np.arange(n_phi) + 0.5) * 2 * np.pi / n_phi
    T, P = np.meshgrid(th, ph, indexing="ij")
    dirs = np.stack([np.sin(T) * np.cos(P), np.sin(T) * np.sin(P), np.cos(T)], -1).reshape(-1, 3).astype(np.float32)
    w = (np.sin(T) * (np.pi / n_theta) * (2 * np.pi / n_phi)).reshape(-1)
    return dirs, w


def evals(o, mat, wo, dirs):
    return np.array([o.bsdf_eval(mat, wo, d, N, DPDU, UV) for d in dirs], np.float64)


def samples(o, mat, wo, n, seed=1):
    u = np.random.default_rng(seed).uniform(size=(n, 3)).astype(np.float32)
    return np.array([o.bsdf_sample(mat, wo, N, DPDU, UV, x) for x in u], np.float64)


@pytest.mark.parametrize("mat", [
    LAMBERT,
    pytest.param(METAL, marks=pytest.mark.xfail(reason=QUIRK_D, strict=True)),
    pytest.param(FLOOR, marks=pytest.mark.xfail(reason=QUIRK_D, strict=True)),
    pytest.param(BRUSHED, marks=pytest.mark.xfail(reason=QUIRK_D, strict=True)),
])
def test_density_integrates_to_one(oracle, mat):
    dirs, w = sphere_grid(256, 512)
    pdf = evals(oracle, mat, WO, dirs)[:, 3]
    assert np.isfinite(pdf).all() and (pdf >= 0).all()
    assert abs(float((pdf * w).sum()) - 1.0) < 2e-2


@pytest.mark.parametrize("mat", [
    LAMBERT, METAL, FLOOR, BRUSHED,
    pytest.param(FROST, marks=pytest.mark.xfail(reason=QUIRK_RD, strict=True)),
])
def test_evaluating_a_sampled_direction_returns_what_the_sampler_returned(oracle, mat):
    s = samples(oracle, mat, WO, 400)
    ok = np.isfinite(s).all(axis=1) & (s[:, 6] > 0)
    assert ok.sum() > 200
    worst = 0.0
    for row in s[ok]:
        e = np.float64(oracle.bsdf_eval(mat, WO, np.float32(row[:3]), N, DPDU, UV))
        worst = max(worst, float(np.max(np.abs(e - row[3:7]) / (np.abs(row[3:7]) + 1e-6))))
    assert worst < 5e-3


@pytest.mark.parametrize("mat", [LAMBERT, METAL, BRUSHED])
def test_reflection_is_reciprocal(oracle, mat):
    rng = np.random.default_rng(5)
    for _ in range(200):
        a, b = direction(*rng.uniform([0.05, 0], [1.4, 6.28])), direction(*rng.uniform([0.05, 0], [1.4, 6.28]))
        f_ab = np.float64(oracle.bsdf_eval(mat, a, b, N, DPDU, UV))[:3]
        f_ba = np.float64(oracle.bsdf_eval(mat, b, a, N, DPDU, UV))[:3]
        assert np.allclose(f_ab, f_ba, rtol=2e-4, atol=1e-7)


def test_no_lobe_creates_energy(oracle):
    """Lambertian: the cosine-weighted integral of fr is the albedo.  Mirror: the sample's weight fr |cos| / pdf is
    the specular colour.  Dielectric: the reflected sample weighs `specular`, the refracted one `specular` times
    the radiance compression (ei/et)^2 of the boundary, and they are chosen with probabilities F and 1 - F."""
    dirs, w = sphere_grid(128, 256)
    e = evals(oracle, LAMBERT, WO, dirs)
    albedo = (e[:, :3] * np.abs(dirs[:, 2:3]) * w[:, None]).sum(axis=0)
    assert np.allclose(albedo, MATS[LAMBERT]["diffuse"], rtol=5e-3)
    s = samples(oracle, MIRROR, WO, 4)
    assert np.allclose(s[:, 3:6] * np.abs(s[:, 2:3]) / s[:, 6:7], 0.9, rtol=1e-5)
    assert np.allclose(s[:, :3], [-WO[0], -WO[1], WO[2]], atol=1e-6)
    s = samples(oracle, GLASS, WO, 2000)
    weight = s[:, 3] * np.abs(s[:, 2]) / s[:, 6]
    refl = s[:, 2] > 0
    assert np.allclose(weight[refl], 1.0, rtol=1e-4) and np.allclose(weight[~refl], (1.0 / 1.5) ** 2, rtol=1e-4)
    # Snell: sin(theta_t) = sin(theta_i) / 1.5 for the refracted directions, and both probabilities seen
    sin_i, sin_t = np.sqrt(1 - WO[2] ** 2), np.sqrt(1 - s[~refl, 2] ** 2)
    assert np.allclose(sin_t, sin_i / 1.5, rtol=1e-4)
    F = s[refl, 6].mean()
    assert 0.03 < F < 0.08 and abs(refl.mean() - F) < 0.02 and np.allclose(s[~refl, 6], 1 - F, rtol=1e-4)


@pytest.mark.parametrize("mat", [
    LAMBERT,
    pytest.param(METAL, marks=pytest.mark.xfail(reason=QUIRK_D, strict=True)),
])
def test_sampled_directions_follow_the_reported_density(oracle, mat):
    """Histogram of 40 000 sampled directions over 8 x 16 (cos theta, phi) cells of the upper hemisphere against the
    reported density integrated over each cell."""
    s = samples(oracle, mat, WO, 40000, seed=9)
    wi = s[np.isfinite(s).all(axis=1) & (s[:, 6] > 0), :3]
    nz, nphi = 8, 16
    iz = np.clip((wi[:, 2] * nz).astype(int), 0, nz - 1)
    ip = np.clip(((np.arctan2(wi[:, 1], wi[:, 0]) % (2 * np.pi)) / (2 * np.pi) * nphi).astype(int), 0, nphi - 1)
    up = wi[:, 2] > 0
    hist = np.zeros((nz, nphi))
    np.add.at(hist, (iz[up], ip[up]), 1.0)
    hist /= len(s)
    # expected mass per cell: the density at 6 x 6 midpoints (cells are equal-area in (z, phi))
    sub = 6
    want = np.zeros((nz, nphi))
    for a in range(nz):
        for b in range(nphi):
            z = (a + (np.arange(sub) + 0.5) / sub) / nz
            ph = (b + (np.arange(sub) + 0.5) / sub) / nphi * 2 * np.pi
            Z, P = np.meshgrid(z, ph, indexing="ij")
            R = np.sqrt(1 - Z * Z)
            d = np.stack([R * np.cos(P), R * np.sin(P), Z], -1).reshape(-1, 3).astype(np.float32)
            want[a, b] = evals(oracle, mat, WO, d)[:, 3].mean() * (2 * np.pi / (nz * nphi))
    assert np.abs(hist - want).max() < 4e-3 and abs(hist.sum() - want.sum()) < 2e-2
