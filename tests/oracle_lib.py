"""ctypes binding of the CPU oracle (oracle/liboracle.so).  Test infrastructure:
only tests, the smoke check and bench.py's cpu_baseline leg load this."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")


class Counts(C.Structure):
    _fields_ = [(n, C.c_uint64 * 4) for n in ("rays", "nodes", "tris", "spheres", "lines", "hits")] + [
        ("samples", C.c_uint64), ("invalid", C.c_uint64), ("bounces", C.c_uint64), ("sphere_box_rejects", C.c_uint64)]

    def as_dict(self):
        d = {n: list(getattr(self, n)) for n in ("rays", "nodes", "tris", "spheres", "lines", "hits")}
        d.update(samples=self.samples, invalid=self.invalid, bounces=self.bounces, sphere_box_rejects=self.sphere_box_rejects)
        return d


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        fp = C.POINTER(C.c_float)
        L.oracle_load.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.oracle_free.argtypes = [C.c_void_p]
        L.oracle_last_error.restype = C.c_char_p
        L.oracle_info.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.oracle_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, fp, C.POINTER(Counts)]
        L.oracle_render_rows.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, fp,
                                         C.POINTER(Counts)]
        L.oracle_tonemap.argtypes = [fp, fp, C.c_int, C.c_int]
        L.oracle_primary_ids.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_int32)]
        L.oracle_primary_ids_bruteforce.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_int32)]
        L.oracle_fps.argtypes = [C.c_int, fp, fp, C.c_int, fp]
        L.oracle_wang_hash.argtypes = [C.c_uint32]
        L.oracle_wang_hash.restype = C.c_uint32
        L.oracle_rng_floats.argtypes = [C.c_uint32, C.c_int, fp]
        L.oracle_rng_raw.argtypes = [C.c_uint32, C.c_int, C.POINTER(C.c_uint32)]
        L.oracle_camera_ray.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, fp]
        L.oracle_trace.argtypes = [C.c_void_p, fp, fp, C.c_float, C.c_float, fp]
        L.oracle_occluded.argtypes = [C.c_void_p, fp, fp, C.c_float, C.c_float]
        L.oracle_sample_camera.argtypes = [C.c_void_p, fp, fp]
        L.oracle_texture.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p]
        L.oracle_bsdf_sample.argtypes = [C.c_void_p, C.c_int, fp, fp, fp, fp, fp, fp]
        L.oracle_bsdf_eval.argtypes = [C.c_void_p, C.c_int, fp, fp, fp, fp, fp, fp]
        L.oracle_light_sample.argtypes = [C.c_void_p, C.c_int, fp, fp, fp]
        L.oracle_sample.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, fp]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f32(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float32))


class Oracle:
    def __init__(self, scene_json):
        self.L = lib()
        h = C.c_void_p()
        if self.L.oracle_load(scene_json.encode(), C.byref(h)) != 0:
            raise RuntimeError(self.L.oracle_last_error().decode())
        self.h = h
        info = (C.c_int * 8)()
        self.L.oracle_info(self.h, info)
        (self.width, self.height, self.integrator, self.n_prims, self.n_nodes, self.n_lights,
         self.n_materials, self.max_depth) = list(info)

    def close(self):
        if self.h:
            self.L.oracle_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, first_iter, n_iter, threads=0, acc=None, counts=None, rows=None):
        """Returns the accumulated sum image (H, W, 3), row 0 = bottom scanline."""
        if threads <= 0:
            threads = min(os.cpu_count() or 1, 64)
        if acc is None:
            acc = np.zeros((self.height, self.width, 3), np.float32)
        y0, y1 = rows if rows is not None else (0, self.height)
        rc = self.L.oracle_render_rows(self.h, first_iter, n_iter, threads, y0, y1, _fp(acc),
                                       C.byref(counts) if counts is not None else None)
        if rc != 0:
            raise ValueError("bad row range")
        return acc

    def tonemap(self, mean, filmic=True):
        mean = _f32(mean)
        out = np.empty_like(mean)
        self.L.oracle_tonemap(_fp(mean), _fp(out), mean.size // 3, 1 if filmic else 0)
        return out

    def primary_ids(self, it=1, bruteforce=False):
        ids = np.empty((self.height, self.width), np.int32)
        fn = self.L.oracle_primary_ids_bruteforce if bruteforce else self.L.oracle_primary_ids
        fn(self.h, it, ids.ctypes.data_as(C.POINTER(C.c_int32)))
        return ids

    def camera_ray(self, x, y, it=1):
        o = np.empty(6, np.float32)
        self.L.oracle_camera_ray(self.h, x, y, it, _fp(o))
        return o

    def trace(self, origin, direction, tmin=1e-3, tmax=np.inf):
        o = np.zeros(12, np.float32)
        pid = self.L.oracle_trace(self.h, _fp(_f32(origin)), _fp(_f32(direction)), tmin, tmax, _fp(o))
        return pid, o

    def occluded(self, origin, direction, tmin=1e-3, tmax=np.inf):
        return bool(self.L.oracle_occluded(self.h, _fp(_f32(origin)), _fp(_f32(direction)), tmin, tmax))

    def texture(self, idx):
        """Texture `idx` as loaded: (h, w, 4) uint8, row 0 = bottom."""
        wh = (C.c_int * 2)()
        if self.L.oracle_texture(self.h, idx, wh, None) != 0:
            raise IndexError(idx)
        out = np.empty((wh[1], wh[0], 4), np.uint8)
        self.L.oracle_texture(self.h, idx, wh, out.ctypes.data_as(C.c_void_p))
        return out

    def sample_camera(self, pos):
        """(raster index or -1, [importance, pdf, tmax])"""
        o = np.zeros(3, np.float32)
        raster = self.L.oracle_sample_camera(self.h, _fp(_f32(pos)), _fp(o))
        return raster, o

    def bsdf_sample(self, mat, wo, n, dpdu, uv, u3):
        o = np.zeros(7, np.float32)
        self.L.oracle_bsdf_sample(self.h, mat, _fp(_f32(wo)), _fp(_f32(n)), _fp(_f32(dpdu)), _fp(_f32(uv)),
                                  _fp(_f32(u3)), _fp(o))
        return o

    def bsdf_eval(self, mat, wo, wi, n, dpdu, uv):
        o = np.zeros(4, np.float32)
        self.L.oracle_bsdf_eval(self.h, mat, _fp(_f32(wo)), _fp(_f32(wi)), _fp(_f32(n)), _fp(_f32(dpdu)),
                                _fp(_f32(uv)), _fp(o))
        return o

    def light_sample(self, idx, pos, u2):
        o = np.zeros(11, np.float32)
        self.L.oracle_light_sample(self.h, idx, _fp(_f32(pos)), _fp(_f32(u2)), _fp(o))
        return o

    def sample(self, pixel, it):
        o = np.zeros(3, np.float32)
        ok = self.L.oracle_sample(self.h, pixel, it, _fp(o))
        return bool(ok), o


FPS_OPS = {"sin": 0, "cos": 1, "tan": 2, "exp": 3, "log": 4, "acos": 5, "asin": 6, "atan": 7, "pow": 8, "rsqrt": 9,
           "fmin": 10, "fmax": 11, "f2i": 12}


def fps(op, x, y=None):
    x = _f32(x)
    out = np.empty_like(x)
    yy = _fp(_f32(y)) if y is not None else None
    lib().oracle_fps(FPS_OPS[op], _fp(x), yy, x.size, _fp(out))
    return out


def wang_hash(v):
    return lib().oracle_wang_hash(v)


def rng_floats(seed, n):
    o = np.empty(n, np.float32)
    lib().oracle_rng_floats(seed, n, _fp(o))
    return o


def rng_raw(seed, n):
    o = np.empty(n, np.uint32)
    lib().oracle_rng_raw(seed, n, o.ctypes.data_as(C.POINTER(C.c_uint32)))
    return o


def save_png(path, img_bottom_up):
    """img: (H, W, 3) float in [0,1], row 0 = bottom; written top row first."""
    from PIL import Image
    a = (np.clip(img_bottom_up[::-1], 0, 1) * 255).astype(np.uint8)
    Image.fromarray(a).save(path)
