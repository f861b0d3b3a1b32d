#!/usr/bin/env python3
"""Deterministic generators for the synthetic benchmark / test scenes.

The reference ships no scene files (its default scene is a path on the
author's disk, CUDA-PathTracer/tracer/CudaRender.h:12), so the five benchmark
configurations are restated here as procedural scenes written in the
reference's JSON schema (CUDA-PathTracer/tracer/parsescene.cpp:46-598):

  cornell_ao   Cornell box, spheres + quads, Lambertian, AO integrator
  cornell_pt   the same box with mirror / glass spheres, path tracing + MIS
  knot_pt      "teapot-full" stand-in: glass torus knot, mirror sphere,
               textured Lambertian ground
  stairs_pt    "staircase2" stand-in: RoughConductor / RoughDielectric /
               Substrate materials, Line railings
  cornell_meshlight  Cornell box lit by 34 emissive triangles
  cornell_smoke  Cornell box with a heterogeneous medium (density grid) in an index-matched box
  cornell_lt / cornell_fog_lt / cornell_smoke_lt  the pt, vpt and smoke boxes under the light-tracing integrator
  cornell_bdpt / cornell_fog_bdpt / cornell_smoke_bdpt  the same three boxes under bidirectional path tracing
  cornell_vpt  Cornell box with a homogeneous medium inside an index-matched
               box boundary and a glass sphere, volumetric path tracing
  car_pt       "bmw-m6" stand-in: >= 1 M triangle displaced body, glass
               canopy, wheels, ground, large quad lights

Everything is a pure function of its arguments (integer lattices + closed
forms, no random state), so the files need not be committed at full size.

usage: gen_scenes.py OUTDIR NAME [--detail D] [--width W --height H] [--spp-depth N]
"""
import argparse
import json
import os
import struct
import sys

import numpy as np


# --------------------------------------------------------------------------
# mesh helpers
# --------------------------------------------------------------------------
class Mesh:
    def __init__(self):
        self.pos, self.nrm, self.uv, self.faces = [], [], [], []
        self._n = 0

    def add(self, pos, nrm, uv, faces):
        pos = np.asarray(pos, np.float32).reshape(-1, 3)
        nrm = np.asarray(nrm, np.float32).reshape(-1, 3)
        uv = np.asarray(uv, np.float32).reshape(-1, 2)
        faces = np.asarray(faces, np.int64).reshape(-1, 3) + self._n
        self.pos.append(pos); self.nrm.append(nrm); self.uv.append(uv); self.faces.append(faces)
        self._n += len(pos)

    def quad(self, p0, p1, p2, p3, n):
        """Quad p0..p3 (counter-clockwise seen from the side n points to)."""
        self.add([p0, p1, p2, p3], [n] * 4, [[0, 0], [1, 0], [1, 1], [0, 1]], [[0, 1, 2], [0, 2, 3]])

    def box(self, centre, half, rot_y_deg=0.0):
        c = np.asarray(centre, np.float64)
        h = np.asarray(half, np.float64)
        a = np.deg2rad(rot_y_deg)
        R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        E = np.eye(3)
        for ax in range(3):
            for sgn in (-1.0, 1.0):
                iu, iv = (ax + 1) % 3, (ax + 2) % 3
                if sgn < 0:
                    iu, iv = iv, iu
                corners = [R @ (sgn * E[ax] * h[ax] + su * E[iu] * h[iu] + sv * E[iv] * h[iv]) + c
                           for su, sv in ((-1, -1), (1, -1), (1, 1), (-1, 1))]
                self.quad(*corners, R @ (sgn * E[ax]))

    def grid_surface(self, P, N, UV, closed_u=False, closed_v=False):
        """Triangulates a (nu, nv) lattice of points with normals."""
        nu, nv = P.shape[:2]
        idx = np.arange(nu * nv).reshape(nu, nv)
        iu = np.arange(nu if closed_u else nu - 1)
        iv = np.arange(nv if closed_v else nv - 1)
        a = idx[np.ix_(iu, iv)]
        b = idx[np.ix_((iu + 1) % nu, iv)]
        c = idx[np.ix_((iu + 1) % nu, (iv + 1) % nv)]
        d = idx[np.ix_(iu, (iv + 1) % nv)]
        f = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([a, c, d], -1).reshape(-1, 3)])
        self.add(P.reshape(-1, 3), N.reshape(-1, 3), UV.reshape(-1, 2), f)

    def arrays(self):
        return (np.concatenate(self.pos), np.concatenate(self.nrm), np.concatenate(self.uv),
                np.concatenate(self.faces))

    def ntris(self):
        return sum(len(f) for f in self.faces)

    def write_obj(self, path):
        pos, nrm, uv, faces = self.arrays()
        with open(path, "w") as f:
            for p in pos:
                f.write("v %.9g %.9g %.9g\n" % tuple(p))
            for t in uv:
                f.write("vt %.9g %.9g\n" % tuple(t))
            for n in nrm:
                f.write("vn %.9g %.9g %.9g\n" % tuple(n))
            for a, b, c in faces + 1:
                f.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (a, a, a, b, b, b, c, c, c))

    def write_ply(self, path):
        pos, nrm, uv, faces = self.arrays()
        v = np.empty(len(pos), dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("nx", "<f4"), ("ny", "<f4"),
                                      ("nz", "<f4"), ("s", "<f4"), ("t", "<f4")])
        v["x"], v["y"], v["z"] = pos[:, 0], pos[:, 1], pos[:, 2]
        v["nx"], v["ny"], v["nz"] = nrm[:, 0], nrm[:, 1], nrm[:, 2]
        v["s"], v["t"] = uv[:, 0], uv[:, 1]
        fa = np.empty(len(faces), dtype=[("n", "u1"), ("a", "<i4"), ("b", "<i4"), ("c", "<i4")])
        fa["n"] = 3
        fa["a"], fa["b"], fa["c"] = faces[:, 0], faces[:, 1], faces[:, 2]
        hdr = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\n"
               "property float x\nproperty float y\nproperty float z\n"
               "property float nx\nproperty float ny\nproperty float nz\n"
               "property float s\nproperty float t\n"
               "element face %d\nproperty list uchar int vertex_indices\nend_header\n") % (len(pos), len(faces))
        with open(path, "wb") as f:
            f.write(hdr.encode())
            f.write(v.tobytes())
            f.write(fa.tobytes())


def normalize(v):
    return v / np.maximum(np.linalg.norm(v, axis=-1, keepdims=True), 1e-30)


def lattice_normals(P, closed_u, closed_v):
    """Smooth normals of a parametric lattice from central differences."""
    def diff(a, axis, closed):
        if closed:
            return np.roll(a, -1, axis) - np.roll(a, 1, axis)
        return np.gradient(a, axis=axis)
    du = diff(P, 0, closed_u)
    dv = diff(P, 1, closed_v)
    return normalize(np.cross(du, dv))


def ellipsoid(centre, radii, nu, nv, displace=None):
    u = np.linspace(0.0, 2 * np.pi, nu, endpoint=False)
    v = np.linspace(0.02, np.pi - 0.02, nv)
    U, V = np.meshgrid(u, v, indexing="ij")
    d = np.stack([np.sin(V) * np.cos(U), np.cos(V), np.sin(V) * np.sin(U)], -1)
    r = np.ones_like(U)
    if displace is not None:
        r = r + displace(U, V)
    P = d * r[..., None] * np.asarray(radii) + np.asarray(centre)
    N = -lattice_normals(P, True, False)
    UV = np.stack([U / (2 * np.pi), V / np.pi], -1)
    m = Mesh()
    m.grid_surface(P.astype(np.float32), N.astype(np.float32), UV.astype(np.float32), closed_u=True)
    return m


def torus(centre, R, r, nu, nv):
    """Torus whose axle is the z axis (a wheel seen from the side of a car that drives along x)."""
    u = np.linspace(0.0, 2 * np.pi, nu, endpoint=False)
    v = np.linspace(0.0, 2 * np.pi, nv, endpoint=False)
    U, V = np.meshgrid(u, v, indexing="ij")
    P = np.stack([(R + r * np.cos(V)) * np.cos(U), (R + r * np.cos(V)) * np.sin(U), r * np.sin(V)], -1)
    N = np.stack([np.cos(V) * np.cos(U), np.cos(V) * np.sin(U), np.sin(V)], -1)
    P = P + np.asarray(centre)
    UV = np.stack([U / (2 * np.pi), V / (2 * np.pi)], -1)
    m = Mesh()
    m.grid_surface(P.astype(np.float32), N.astype(np.float32), UV.astype(np.float32), closed_u=True, closed_v=True)
    return m


def torus_knot(centre, scale, p, q, nu, nv, tube):
    t = np.linspace(0.0, 2 * np.pi, nu, endpoint=False)
    def curve(t):
        r = 2.0 + np.cos(q * t)
        return np.stack([r * np.cos(p * t), np.sin(q * t) * 1.2, r * np.sin(p * t)], -1)
    C = curve(t)
    h = 1e-4
    T = normalize(curve(t + h) - curve(t - h))
    A = normalize(curve(t + h) - 2 * C + curve(t - h) + 1e-9)
    B = normalize(np.cross(T, A))
    Nn = np.cross(B, T)
    v = np.linspace(0.0, 2 * np.pi, nv, endpoint=False)
    cv, sv = np.cos(v), np.sin(v)
    D = Nn[:, None, :] * cv[None, :, None] + B[:, None, :] * sv[None, :, None]
    P = (C[:, None, :] + tube * D) * scale + np.asarray(centre)
    UV = np.stack(np.meshgrid(t / (2 * np.pi) * 16, v / (2 * np.pi), indexing="ij"), -1)
    m = Mesh()
    m.grid_surface(P.astype(np.float32), D.astype(np.float32), UV.astype(np.float32), closed_u=True, closed_v=True)
    return m


def heightfield(x0, x1, z0, z1, nx, nz, fn):
    x = np.linspace(x0, x1, nx)
    z = np.linspace(z0, z1, nz)
    X, Z = np.meshgrid(x, z, indexing="ij")
    Y = fn(X, Z)
    P = np.stack([X, Y, Z], -1)
    N = lattice_normals(P, False, False)
    N = np.where(N[..., 1:2] < 0, -N, N)
    UV = np.stack([(X - x0) / (x1 - x0), (Z - z0) / (z1 - z0)], -1)
    m = Mesh()
    # wind so that the geometric normal points up (+y)
    m.grid_surface(P[:, ::-1].astype(np.float32), N[:, ::-1].astype(np.float32), UV[:, ::-1].astype(np.float32))
    return m


def write_checker_ppm(path, size=256, cells=8, c0=(200, 200, 200), c1=(60, 70, 90)):
    y, x = np.mgrid[0:size, 0:size]
    k = ((x * cells // size) + (y * cells // size)) & 1
    img = np.where(k[..., None] == 0, np.array(c0, np.uint8), np.array(c1, np.uint8)).astype(np.uint8)
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (size, size))
        f.write(img.tobytes())


def write_sky_pfm(path, w=64, h=32):
    """Small analytic sky: bright band near the horizon plus a warm sun blob."""
    v, u = np.mgrid[0:h, 0:w]
    theta = (v + 0.5) / h * np.pi
    phi = (u + 0.5) / w * 2 * np.pi
    sky = 0.35 + 0.65 * np.clip(np.cos(theta), 0, 1)[..., None] * np.array([0.5, 0.7, 1.0])
    sun = np.exp(-((theta - 0.9) ** 2 + (phi - 1.2) ** 2) * 12.0)[..., None] * np.array([12.0, 10.0, 7.0])
    img = (sky + sun).astype("<f4")
    with open(path, "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (w, h))
        f.write(img[::-1].tobytes())   # PFM stores the bottom row first


# --------------------------------------------------------------------------
# scenes
# --------------------------------------------------------------------------
def cornell_meshes(out):
    room = Mesh()
    room.quad([-1, 0, 1], [1, 0, 1], [1, 0, -1], [-1, 0, -1], [0, 1, 0])       # floor
    room.quad([-1, 2, -1], [1, 2, -1], [1, 2, 1], [-1, 2, 1], [0, -1, 0])      # ceiling
    room.quad([-1, 0, -1], [1, 0, -1], [1, 2, -1], [-1, 2, -1], [0, 0, 1])     # back
    room.write_obj(os.path.join(out, "room.obj"))
    left = Mesh(); left.quad([-1, 0, 1], [-1, 0, -1], [-1, 2, -1], [-1, 2, 1], [1, 0, 0])
    left.write_obj(os.path.join(out, "left.obj"))
    right = Mesh(); right.quad([1, 0, -1], [1, 0, 1], [1, 2, 1], [1, 2, -1], [-1, 0, 0])
    right.write_obj(os.path.join(out, "right.obj"))
    short = Mesh(); short.box([0.33, 0.3, 0.37], [0.3, 0.3, 0.3], -17.0)
    short.write_obj(os.path.join(out, "short.obj"))
    tall = Mesh(); tall.box([-0.35, 0.6, -0.3], [0.3, 0.6, 0.3], 18.0)
    tall.write_obj(os.path.join(out, "tall.obj"))
    light = Mesh(); light.quad([-0.24, 1.98, -0.22], [0.23, 1.98, -0.22], [0.23, 1.98, 0.16], [-0.24, 1.98, 0.16], [0, -1, 0])
    light.write_obj(os.path.join(out, "light.obj"))


CORNELL_CAM = {"position": [0, 1, 6.8], "lookat": [0, 1, 0], "up": [0, 1, 0], "fov": 19.5}
CORNELL_MATS = [
    {"name": "white", "bsdf": "lambertian", "diffuse": [0.725, 0.71, 0.68]},
    {"name": "red", "bsdf": "lambertian", "diffuse": [0.63, 0.065, 0.05]},
    {"name": "green", "bsdf": "lambertian", "diffuse": [0.14, 0.45, 0.091]},
    {"name": "matte", "bsdf": "lambertian", "diffuse": [0.78, 0.78, 0.78]},
]


def scene_cornell_ao(out, a):
    cornell_meshes(out)
    return {
        "screen_width": a.width or 512, "screen_height": a.height or 512, "epsilon": 0.001,
        "integrator": "ao", "maxDist": 1.0,
        "camera": dict(CORNELL_CAM),
        "material": CORNELL_MATS,
        "scene": [
            {"mesh": "room.obj", "material": "white"},
            {"mesh": "left.obj", "material": "red"},
            {"mesh": "right.obj", "material": "green"},
            {"mesh": "short.obj", "material": "white"},
            {"mesh": "tall.obj", "material": "white"},
            {"sphere": True, "center": [0.45, 0.85, 0.35], "radius": 0.25, "material": "white"},
            {"sphere": True, "center": [-0.55, 0.2, 0.55], "radius": 0.2, "material": "white"},
        ],
        "light": [{"mesh": "light.obj", "material": "matte", "radiance": [17, 12, 4]}],
    }


def scene_cornell_pt(out, a):
    s = scene_cornell_ao(out, a)
    s["integrator"] = "pt"
    del s["maxDist"]
    s["maxDepth"] = a.depth or 8
    s["material"] = CORNELL_MATS + [
        {"name": "mirror", "bsdf": "mirror"},
        {"name": "glass", "bsdf": "dielectric", "insideIOR": 1.5, "outsideIOR": 1.0},
    ]
    s["scene"][5]["material"] = "glass"
    s["scene"][6]["material"] = "mirror"
    return s


def scene_cornell_meshlight(out, a):
    """Cornell box lit by a tessellated panel (32 emissive triangles: one area light each, a long
    light distribution) and a small second panel on the back wall."""
    s = scene_cornell_pt(out, a)
    panel = Mesh()
    n = 4
    for i in range(n):
        for j in range(n):
            x0, x1 = -0.4 + 0.8 * i / n, -0.4 + 0.8 * (i + 1) / n
            z0, z1 = -0.35 + 0.6 * j / n, -0.35 + 0.6 * (j + 1) / n
            panel.quad([x0, 1.98, z0], [x1, 1.98, z0], [x1, 1.98, z1], [x0, 1.98, z1], [0, -1, 0])
    panel.write_obj(os.path.join(out, "panel.obj"))
    spot = Mesh(); spot.quad([-0.2, 1.2, -0.98], [0.2, 1.2, -0.98], [0.2, 1.5, -0.98], [-0.2, 1.5, -0.98], [0, 0, 1])
    spot.write_obj(os.path.join(out, "spot.obj"))
    s["light"] = [{"mesh": "panel.obj", "material": "matte", "radiance": [9, 7, 3]},
                  {"mesh": "spot.obj", "material": "matte", "radiance": [2, 4, 9]}]
    return s


def scene_cornell_vpt(out, a):
    cornell_meshes(out)
    fog = Mesh(); fog.box([0.0, 0.8, 0.0], [0.75, 0.55, 0.75], 0.0)
    fog.write_obj(os.path.join(out, "fogbox.obj"))
    return {
        "screen_width": a.width or 1024, "screen_height": a.height or 1024, "epsilon": 0.001,
        "integrator": "vpt", "maxDepth": a.depth or 16,
        "camera": dict(CORNELL_CAM),
        "medium": [{"name": "fog", "type": "homogeneous", "sigmaA": [0.1, 0.1, 0.1], "sigmaS": [1, 1, 1],
                    "scale": 1.5, "g": 0}],
        "material": CORNELL_MATS + [{"name": "glass", "bsdf": "dielectric", "insideIOR": 1.5, "outsideIOR": 1.0}],
        "scene": [
            {"mesh": "room.obj", "material": "white"},
            {"mesh": "left.obj", "material": "red"},
            {"mesh": "right.obj", "material": "green"},
            {"mesh": "fogbox.obj", "inside": "fog", "outside": ""},
            {"sphere": True, "center": [-0.1, 0.8, 0.1], "radius": 0.3, "material": "glass", "inside": "fog",
             "outside": "fog"},
            {"sphere": True, "center": [0.6, 0.2, 0.75], "radius": 0.2, "material": "white"},
        ],
        "light": [{"mesh": "light.obj", "material": "matte", "radiance": [17, 12, 4]}],
    }


def scene_cornell_smoke(out, a):
    """Cornell box with a plume of smoke: a heterogeneous medium (density grid, delta / ratio
    tracking) inside an index-matched boundary box, g != 0 so that the Henyey-Greenstein
    branches of the phase function run too."""
    s = scene_cornell_vpt(out, a)
    n = 20
    z, y, x = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    u, v, w = (x + 0.5) / n - 0.5, (y + 0.5) / n, (z + 0.5) / n - 0.5
    r = np.sqrt((u - 0.12 * np.sin(7 * v)) ** 2 + (w - 0.1 * np.cos(5 * v)) ** 2)
    dens = np.clip(1.0 - r / (0.12 + 0.3 * v), 0, 1) * (0.35 + 0.65 * np.abs(np.sin(9 * u + 4 * v) * np.cos(8 * w)))
    with open(os.path.join(out, "smoke.txt"), "w") as f:
        for val in dens.reshape(-1):   # z-major, then y, then x
            f.write("%.6f\n" % val)
    s["medium"] = [{"name": "fog", "type": "heterogeneous", "sigmaA": [0.5, 0.5, 0.5], "sigmaS": [5.5, 5.5, 5.5],
                    "scale": 2.0, "g": 0.4, "nx": n, "ny": n, "nz": n, "p0": [-0.75, 0.25, -0.75],
                    "p1": [0.75, 1.35, 0.75], "density": "smoke.txt", "iterMax": 200}]
    s["scene"][4] = {"sphere": True, "center": [0.55, 0.3, 0.7], "radius": 0.3, "material": "glass"}
    return s


def scene_cornell_lt(out, a):
    """The path-traced Cornell box (mirror and glass spheres), rendered by light tracing."""
    s = scene_cornell_pt(out, a)
    s["integrator"] = "lt"
    return s


def scene_cornell_fog_lt(out, a):
    """Light tracing through the homogeneous fog of cornell_vpt: medium vertices, index-matched
    boundaries and transmittance on the camera connections."""
    s = scene_cornell_vpt(out, a)
    s["integrator"] = "lt"
    s["screen_width"], s["screen_height"] = a.width or 512, a.height or 512
    return s


def scene_cornell_smoke_lt(out, a):
    """Light tracing through the density grid of cornell_smoke (transmittance draws random numbers)."""
    s = scene_cornell_smoke(out, a)
    s["integrator"] = "lt"
    s["screen_width"], s["screen_height"] = a.width or 512, a.height or 512
    return s


def scene_cornell_bdpt(out, a):
    """The path-traced Cornell box (mirror and glass spheres), rendered by bidirectional path tracing."""
    s = scene_cornell_pt(out, a)
    s["integrator"] = "bdpt"
    return s


def scene_cornell_fog_bdpt(out, a):
    """BDPT through the homogeneous fog of cornell_vpt: medium vertices on both sub-paths,
    index-matched boundaries, transmittance on every connection."""
    s = scene_cornell_vpt(out, a)
    s["integrator"] = "bdpt"
    s["screen_width"], s["screen_height"] = a.width or 512, a.height or 512
    return s


def scene_cornell_smoke_bdpt(out, a):
    """BDPT through the density grid of cornell_smoke (connections draw random numbers)."""
    s = scene_cornell_smoke(out, a)
    s["integrator"] = "bdpt"
    s["screen_width"], s["screen_height"] = a.width or 512, a.height or 512
    return s


def scene_knot_pt(out, a):
    d = a.detail
    nu, nv = max(24, int(600 * d)), max(8, int(128 * d))
    torus_knot([0, 1.15, 0], 0.33, 2, 3, nu, nv, 0.42).write_ply(os.path.join(out, "knot.ply"))
    g = Mesh(); g.quad([-6, 0, 6], [6, 0, 6], [6, 0, -6], [-6, 0, -6], [0, 1, 0])
    g.write_obj(os.path.join(out, "ground.obj"))
    wall = Mesh(); wall.quad([-6, 0, -3], [6, 0, -3], [6, 6, -3], [-6, 6, -3], [0, 0, 1])
    wall.write_obj(os.path.join(out, "wall.obj"))
    L = Mesh(); L.quad([-1.5, 4.5, -1.0], [1.5, 4.5, -1.0], [1.5, 4.5, 1.5], [-1.5, 4.5, 1.5], [0, -1, 0])
    L.write_obj(os.path.join(out, "light.obj"))
    write_checker_ppm(os.path.join(out, "checker.ppm"))
    return {
        "screen_width": a.width or 1024, "screen_height": a.height or 1024, "epsilon": 0.001,
        "integrator": "pt", "maxDepth": a.depth or 8,
        "camera": {"position": [0.5, 2.2, 5.5], "lookat": [0, 1.0, 0], "up": [0, 1, 0], "fov": 36},
        "material": [
            {"name": "checker", "bsdf": "lambertian", "diffuse": "checker.ppm"},
            {"name": "wall", "bsdf": "lambertian", "diffuse": [0.6, 0.6, 0.65]},
            {"name": "glass", "bsdf": "dielectric", "insideIOR": 1.5, "outsideIOR": 1.0},
            {"name": "mirror", "bsdf": "mirror", "specular": [0.9, 0.9, 0.9]},
            {"name": "matte", "bsdf": "lambertian", "diffuse": [0.8, 0.8, 0.8]},
        ],
        "scene": [
            {"mesh": "ground.obj", "material": "checker"},
            {"mesh": "wall.obj", "material": "wall"},
            {"mesh": "knot.ply", "material": "glass"},
            {"sphere": True, "center": [-2.0, 0.6, -0.8], "radius": 0.6, "material": "mirror"},
            {"sphere": True, "center": [2.0, 0.45, 0.4], "radius": 0.45, "material": "matte"},
        ],
        "light": [{"mesh": "light.obj", "material": "matte", "radiance": [14, 13, 12]}],
    }


def scene_stairs_pt(out, a):
    d = a.detail
    steps = Mesh()
    nsteps = 12
    for i in range(nsteps):
        steps.box([0.0, 0.1 + 0.2 * i, 2.0 - 0.35 * i], [1.2, 0.1, 0.175], 0.0)
    steps.write_obj(os.path.join(out, "steps.obj"))
    n = max(8, int(390 * d))
    heightfield(-4, 4, -4, 4, n, n, lambda X, Z: -0.02 + 0.015 * np.sin(9 * X) * np.cos(7 * Z)).write_ply(
        os.path.join(out, "floor.ply"))
    walls = Mesh()
    walls.quad([-4, -0.1, -4], [4, -0.1, -4], [4, 5, -4], [-4, 5, -4], [0, 0, 1])
    walls.quad([-4, -0.1, 4], [-4, -0.1, -4], [-4, 5, -4], [-4, 5, 4], [1, 0, 0])
    walls.quad([4, -0.1, -4], [4, -0.1, 4], [4, 5, 4], [4, 5, -4], [-1, 0, 0])
    walls.quad([-4, 5, -4], [4, 5, -4], [4, 5, 4], [-4, 5, 4], [0, -1, 0])
    walls.write_obj(os.path.join(out, "walls.obj"))
    pane = Mesh(); pane.box([2.2, 1.4, 0.5], [0.03, 1.2, 1.2], 0.0)
    pane.write_obj(os.path.join(out, "pane.obj"))
    ne = max(8, int(96 * d))
    ellipsoid([-2.2, 0.9, 0.6], [0.7, 0.9, 0.7], 2 * ne, ne).write_ply(os.path.join(out, "vase.ply"))
    win = Mesh()
    win.quad([-3.95, 1.5, -1.5], [-3.95, 1.5, 1.5], [-3.95, 4.0, 1.5], [-3.95, 4.0, -1.5], [1, 0, 0])
    win.quad([-1.5, 4.95, -2.0], [1.5, 4.95, -2.0], [1.5, 4.95, 0.0], [-1.5, 4.95, 0.0], [0, -1, 0])
    win.write_obj(os.path.join(out, "windows.obj"))
    write_checker_ppm(os.path.join(out, "wallpaper.ppm"), 256, 16, (205, 190, 170), (170, 150, 130))
    shapes = [
        {"mesh": "floor.ply", "material": "floor"},
        {"mesh": "walls.obj", "material": "wallpaper"},
        {"mesh": "steps.obj", "material": "wood"},
        {"mesh": "pane.obj", "material": "frost"},
        {"mesh": "vase.ply", "material": "copper"},
        {"sphere": True, "center": [1.0, 0.5, 2.6], "radius": 0.5, "material": "steel"},
    ]
    for i in range(nsteps):   # railing: posts and a hand rail as Line primitives
        x, y, z = 1.25, 0.2 + 0.2 * i, 2.0 - 0.35 * i
        shapes.append({"line": True, "p0": [x, y, z], "p1": [x, y + 0.9, z], "width0": 0.02, "width1": 0.02,
                       "material": "steel"})
        if i + 1 < nsteps:
            shapes.append({"line": True, "p0": [x, y + 0.9, z], "p1": [x, y + 1.1, z - 0.35], "width0": 0.03,
                           "width1": 0.03, "material": "steel"})
    return {
        "screen_width": a.width or 1920, "screen_height": a.height or 1080, "epsilon": 0.001,
        "integrator": "pt", "maxDepth": a.depth or 8,
        "camera": {"position": [0.3, 2.0, 7.5], "lookat": [0, 1.4, 0], "up": [0, 1, 0], "fov": 40},
        "material": [
            {"name": "floor", "bsdf": "substrate", "alpha": 0.05, "diffuse": [0.4, 0.3, 0.2],
             "specular": [0.04, 0.04, 0.04]},
            {"name": "wood", "bsdf": "substrate", "alphaU": 0.1, "alphaV": 0.1, "diffuse": [0.45, 0.28, 0.15],
             "specular": [0.06, 0.06, 0.06]},
            {"name": "wallpaper", "bsdf": "lambertian", "diffuse": "wallpaper.ppm"},
            {"name": "copper", "bsdf": "roughconduct", "alpha": 0.1, "eta": [0.2, 0.92, 1.1], "k": [3.9, 2.45, 2.14]},
            {"name": "steel", "bsdf": "roughconduct", "alpha": 0.2, "eta": [2.5, 2.5, 2.4], "k": [3.1, 3.0, 2.9]},
            {"name": "frost", "bsdf": "roughdielectric", "alpha": 0.05, "insideIOR": 1.5, "outsideIOR": 1.0},
            {"name": "matte", "bsdf": "lambertian", "diffuse": [0.8, 0.8, 0.8]},
        ],
        "scene": shapes,
        "light": [{"mesh": "windows.obj", "material": "matte", "radiance": [9, 9, 10]}],
    }


def scene_car_pt(out, a):
    d = a.detail
    nu, nv = max(32, int(1024 * np.sqrt(d))), max(16, int(512 * np.sqrt(d)))

    def body_disp(U, V):   # panel seams + fine ripples: keeps the BVH honest
        return (0.02 * np.sin(6 * U) * np.sin(5 * V) + 0.006 * np.sin(40 * U + 3 * np.sin(9 * V))
                + 0.004 * np.cos(55 * V) * np.sin(31 * U))
    ellipsoid([0, 0.75, 0], [2.3, 0.62, 1.0], nu, nv, body_disp).write_ply(os.path.join(out, "body.ply"))
    nc = max(16, int(256 * np.sqrt(d)))
    ellipsoid([-0.2, 1.2, 0], [1.0, 0.5, 0.8], nc, nc // 2).write_ply(os.path.join(out, "canopy.ply"))
    nw = max(12, int(128 * np.sqrt(d)))
    wheels = Mesh()
    for x in (-1.4, 1.4):
        for z in (-0.95, 0.95):
            wheels.add(*torus([x, 0.36, z], 0.24, 0.12, nw, nw).arrays())
    wheels.write_ply(os.path.join(out, "wheels.ply"))
    ng = max(8, int(256 * np.sqrt(d)))
    heightfield(-9, 9, -9, 9, ng, ng, lambda X, Z: 0.004 * np.sin(3 * X) * np.sin(3 * Z)).write_ply(
        os.path.join(out, "ground.ply"))
    lights = Mesh()
    lights.quad([-3, 4.5, -2], [3, 4.5, -2], [3, 4.5, 2], [-3, 4.5, 2], [0, -1, 0])
    lights.quad([-5.5, 0.5, 3.5], [-5.5, 0.5, -3.5], [-5.5, 3.5, -3.5], [-5.5, 3.5, 3.5], [1, 0, 0])
    lights.write_obj(os.path.join(out, "lights.obj"))
    write_checker_ppm(os.path.join(out, "asphalt.ppm"), 256, 32, (120, 120, 125), (90, 90, 96))
    write_sky_pfm(os.path.join(out, "sky.pfm"))
    s = {
        "screen_width": a.width or 1920, "screen_height": a.height or 1080, "epsilon": 0.001,
        "integrator": "pt", "maxDepth": a.depth or 8,
        "camera": {"position": [4.2, 2.1, 5.2], "lookat": [0, 0.7, 0], "up": [0, 1, 0], "fov": 34},
        "material": [
            {"name": "paint", "bsdf": "roughconduct", "alpha": 0.12, "eta": [0.27, 0.68, 1.2], "k": [3.6, 2.6, 2.3],
             "specular": [0.95, 0.35, 0.3]},
            {"name": "glass", "bsdf": "dielectric", "insideIOR": 1.5, "outsideIOR": 1.0},
            {"name": "rubber", "bsdf": "substrate", "alpha": 0.3, "diffuse": [0.03, 0.03, 0.03],
             "specular": [0.04, 0.04, 0.04]},
            {"name": "asphalt", "bsdf": "lambertian", "diffuse": "asphalt.ppm"},
            {"name": "matte", "bsdf": "lambertian", "diffuse": [0.8, 0.8, 0.8]},
        ],
        "scene": [
            {"mesh": "ground.ply", "material": "asphalt"},
            {"mesh": "body.ply", "material": "paint"},
            {"mesh": "canopy.ply", "material": "glass"},
            {"mesh": "wheels.ply", "material": "rubber"},
        ],
        "light": [{"mesh": "lights.obj", "material": "matte", "radiance": [10, 10, 10]}],
    }
    if a.env:
        s["light"].append({"infinite": "sky.pfm", "rotate": [0, 30, 0]})
    return s


SCENES = {
    "cornell_ao": scene_cornell_ao, "cornell_pt": scene_cornell_pt, "cornell_vpt": scene_cornell_vpt,
    "cornell_smoke": scene_cornell_smoke, "cornell_meshlight": scene_cornell_meshlight,
    "cornell_lt": scene_cornell_lt, "cornell_fog_lt": scene_cornell_fog_lt, "cornell_smoke_lt": scene_cornell_smoke_lt,
    "cornell_bdpt": scene_cornell_bdpt, "cornell_fog_bdpt": scene_cornell_fog_bdpt,
    "cornell_smoke_bdpt": scene_cornell_smoke_bdpt,
    "knot_pt": scene_knot_pt, "stairs_pt": scene_stairs_pt, "car_pt": scene_car_pt,
}


def generate(outdir, name, detail=1.0, width=0, height=0, depth=0, env=False):
    os.makedirs(outdir, exist_ok=True)
    a = argparse.Namespace(detail=detail, width=width, height=height, depth=depth, env=env)
    s = SCENES[name](outdir, a)
    path = os.path.join(outdir, "scene.json")
    with open(path, "w") as f:
        json.dump(s, f, indent=1)
    return path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("outdir")
    ap.add_argument("name", choices=sorted(SCENES))
    ap.add_argument("--detail", type=float, default=1.0)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--depth", type=int, default=0)
    ap.add_argument("--env", action="store_true")
    a = ap.parse_args()
    print(generate(a.outdir, a.name, a.detail, a.width, a.height, a.depth, a.env))


if __name__ == "__main__":
    main()
