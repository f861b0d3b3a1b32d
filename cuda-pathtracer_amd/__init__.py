"""Host-side mirror of the reference's render interface over libcaster.so.

The reference drives its device code through three free functions,
``BeginRender / Render / EndRender`` (CUDA-PathTracer/tracer/pathtracer.h:8-10),
wrapped by the ``CudaRender`` singleton (CUDA-PathTracer/tracer/CudaRender.h:9-19:
``InitCudaScene, renderImage, GetRenderReverseImage, GetConfig, SaveImage,
GetIteration``).  This module offers the same names and argument meaning over the
C-ABI of ``include/caster.h``; all rendering happens in the HIP library, and
importing a renderer fails loudly if that library has not been built.

Images are numpy arrays of shape (H, W, 3), float32, row 0 = bottom scanline as
the reference's kernels write them; ``GetRenderReverseImage`` returns the
top-row-first flip the reference hands to its window.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CASTER_LIB selects another build of the same library (tuning experiments)
LIB_PATH = os.environ.get("CASTER_LIB") or os.path.join(_HERE, "libcaster.so")

IT_AO, IT_PT, IT_VPT = 0, 1, 2
INTEGRATOR_NAMES = {0: "ao", 1: "pt", 2: "vpt", 3: "lt", 4: "bdpt"}


class CasterError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("caster error %d: %s" % (code, message))
        self.code = code


class SceneInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "width", "height", "integrator", "max_depth", "n_prims", "n_nodes", "n_lights", "n_materials",
        "n_media", "n_textures", "has_environment", "filmic")] + [("max_dist", C.c_float), ("epsilon", C.c_float)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64 * 4), ("samples", C.c_uint64), ("invalid", C.c_uint64),
                ("stack_overflow", C.c_uint64)]

    def as_dict(self):
        return {"rays": list(self.rays), "samples": self.samples, "invalid": self.invalid,
                "stack_overflow": self.stack_overflow}


STATIC_MAX_RAYS = 1 << 18   # CASTER_STATIC_MAX_RAYS of include/caster.h

KERNEL_CLASSES = ["generate", "extend", "shade", "shadow", "probe", "finish"]


class KernelTimes(C.Structure):
    _fields_ = [("ms", C.c_double * 6), ("launches", C.c_uint64 * 6)]


_lib = None

# every symbol include/caster.h declares
EXPORTS = [
    "caster_scene_load", "caster_scene_load_device", "caster_scene_get_bvh", "caster_scene_free", "caster_scene_get_info", "caster_begin", "caster_begin_streams", "caster_begin_multi",
    "caster_set_rows", "caster_set_row_stripes", "caster_set_lookahead",
    "caster_render", "caster_render_async", "caster_sync", "caster_read_film", "caster_read_film_device",
    "caster_film_add_device", "caster_primary_ids", "caster_trace_rays", "caster_trace_rays_static", "caster_eval_function", "caster_get_stats", "caster_get_walk_counts", "caster_reset_stats",
    "caster_iterations", "caster_end", "caster_last_error", "caster_set_profiling", "caster_get_kernel_times",
    "caster_write_image", "caster_save_image", "caster_iters_in_flight", "caster_checkpoint_save", "caster_checkpoint_load",
]


def _share_hip_runtime_with_torch():
    """PyTorch wheels bundle their own libamdhip64.so.7; libcaster.so asks for the same
    soname.  A process must hold ONE HIP runtime (two copies each open the GPU and the
    second finds none), so when torch is installed its copy is loaded first, before
    libcaster.so resolves the soname.  torch itself is not imported."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Loads libcaster.so (no fallback: the HIP library is the only implementation)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: build it with `make -C cuda-pathtracer_amd/csrc` "
                              "(or __graft_entry__.build())" % LIB_PATH)
        _share_hip_runtime_with_torch()
        L = C.CDLL(LIB_PATH)
        vp, fp = C.c_void_p, C.POINTER(C.c_float)
        L.caster_scene_load.argtypes = [C.c_char_p, C.POINTER(vp)]
        L.caster_scene_load_device.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
        L.caster_scene_get_bvh.argtypes = [vp, fp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.caster_scene_free.argtypes = [vp]
        L.caster_scene_free.restype = None
        L.caster_scene_get_info.argtypes = [vp, C.POINTER(SceneInfo)]
        L.caster_begin.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp)]
        L.caster_begin_streams.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
        L.caster_begin_multi.argtypes = [vp, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
        L.caster_set_rows.argtypes = [vp, C.c_int, C.c_int]
        L.caster_set_lookahead.argtypes = [vp, C.c_int]
        L.caster_set_row_stripes.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]
        L.caster_render.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_int]
        L.caster_render_async.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_int]
        L.caster_sync.argtypes = [vp]
        L.caster_read_film.argtypes = [vp, fp, C.c_int]
        L.caster_read_film_device.argtypes = [vp, vp, C.c_int]
        L.caster_film_add_device.argtypes = [vp, vp, C.c_uint32]
        L.caster_write_image.argtypes = [C.c_char_p, C.c_int, C.c_int, fp]
        L.caster_save_image.argtypes = [vp, C.c_char_p, C.c_int]
        L.caster_checkpoint_save.argtypes = [vp, C.c_char_p]
        L.caster_checkpoint_load.argtypes = [vp, C.c_char_p]
        L.caster_primary_ids.argtypes = [vp, C.c_uint32, C.POINTER(C.c_int32)]
        L.caster_trace_rays.argtypes = [vp, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                        C.POINTER(C.c_int32), C.POINTER(C.c_float)]
        L.caster_trace_rays_static.argtypes = [vp, C.c_uint32, vp, vp, vp, vp, vp]
        L.caster_eval_function.argtypes = [vp, C.c_int, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.caster_get_stats.argtypes = [vp, C.POINTER(Stats)]
        L.caster_get_walk_counts.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.caster_reset_stats.argtypes = [vp]
        L.caster_set_profiling.argtypes = [vp, C.c_int]
        L.caster_get_kernel_times.argtypes = [vp, C.POINTER(KernelTimes)]
        L.caster_iters_in_flight.argtypes = [vp]
        L.caster_iterations.argtypes = [vp]
        L.caster_iterations.restype = C.c_uint32
        L.caster_end.argtypes = [vp]
        L.caster_end.restype = None
        L.caster_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise CasterError(rc, lib().caster_last_error().decode(errors="replace"))


class Scene:
    """A parsed scene file plus its BVH (the reference's ``Scene`` + ``GlobalConfig``
    after ``InitScene``, CUDA-PathTracer/tracer/parsescene.cpp:600-617)."""

    def __init__(self, json_path, build_device=None):
        """`build_device`: HIP device that builds the BVH (None: the host builds it)."""
        h = C.c_void_p()
        if build_device is None:
            _check(lib().caster_scene_load(os.fspath(json_path).encode(), C.byref(h)))
        else:
            _check(lib().caster_scene_load_device(os.fspath(json_path).encode(), int(build_device), C.byref(h)))
        self._h = h
        info = SceneInfo()
        _check(lib().caster_scene_get_info(self._h, C.byref(info)))
        self.info = info
        self.width, self.height = info.width, info.height
        self.integrator = info.integrator

    def bvh(self):
        """(boxes (n_nodes, 6), meta (n_nodes, 4), primitive push ids (n_prims,)) of the binary tree."""
        boxes = np.empty((self.info.n_nodes, 6), np.float32)
        meta = np.empty((self.info.n_nodes, 4), np.int32)
        ids = np.empty(self.info.n_prims, np.int32)
        _check(lib().caster_scene_get_bvh(self._h, boxes.ctypes.data_as(C.POINTER(C.c_float)),
                                          meta.ctypes.data_as(C.POINTER(C.c_int32)),
                                          ids.ctypes.data_as(C.POINTER(C.c_int32))))
        return boxes, meta, ids

    def close(self):
        if getattr(self, "_h", None):
            lib().caster_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Renderer:
    """Device-side render state for one scene on one GPU."""

    def __init__(self, scene, device=0, iters_in_flight=0, n_streams=0, shard="tile"):
        """`device`: a HIP device index, or a list of them for one renderer over several GPUs
        (caster_begin_multi; `shard` = "tile" or "spp" says how the image is dealt to them)."""
        h = C.c_void_p()
        if isinstance(device, (list, tuple)):
            ids = (C.c_int * len(device))(*[int(d) for d in device])
            _check(lib().caster_begin_multi(scene._h, ids, len(device), int(iters_in_flight), int(n_streams),
                                            {"tile": 0, "spp": 1}[shard], C.byref(h)))
        else:
            _check(lib().caster_begin_streams(scene._h, int(device), int(iters_in_flight), int(n_streams), C.byref(h)))
        self._h = h
        self.width, self.height = scene.width, scene.height
        self.integrator = scene.integrator
        self.device = device
        self.iters_in_flight = lib().caster_iters_in_flight(self._h)

    def set_rows(self, y0, y1):
        _check(lib().caster_set_rows(self._h, y0, y1))

    def set_row_stripes(self, y0, y1, rows, stride):
        """Scanlines y0 <= y < y1 with (y - y0) % stride < rows (see sharding.row_stripes)."""
        _check(lib().caster_set_row_stripes(self._h, y0, y1, rows, stride))

    def render(self, first_iter, n_iter=1, reset=False):
        _check(lib().caster_render(self._h, first_iter, n_iter, 1 if reset else 0))

    def set_lookahead(self, n_iter):
        """One-iteration calls trace ``n_iter`` iterations together and add them to the film one call at a time."""
        _check(lib().caster_set_lookahead(self._h, n_iter))

    def render_async(self, first_iter, n_iter=1, reset=False):
        _check(lib().caster_render_async(self._h, first_iter, n_iter, 1 if reset else 0))

    def sync(self):
        _check(lib().caster_sync(self._h))

    def read_film(self, mode=0):
        out = np.empty((self.height, self.width, 3), np.float32)
        _check(lib().caster_read_film(self._h, out.ctypes.data_as(C.POINTER(C.c_float)), mode))
        return out

    def read_film_device(self, device_ptr, mode=2):
        _check(lib().caster_read_film_device(self._h, C.c_void_p(device_ptr), mode))

    def film_add_device(self, device_ptr, n_iters):
        _check(lib().caster_film_add_device(self._h, C.c_void_p(device_ptr), n_iters))

    def save_image(self, path, mode=-1):
        """Writes the film as .png / .exr / .pfm (the reference's SaveImage, CudaRender.cpp:84-90)."""
        _check(lib().caster_save_image(self._h, os.fspath(path).encode(), mode))

    def save_checkpoint(self, path):
        """Writes the raw film sum and the iteration count (caster_checkpoint_save)."""
        _check(lib().caster_checkpoint_save(self._h, os.fspath(path).encode()))

    def load_checkpoint(self, path):
        """Replaces film and iteration count with a checkpoint's; returns the iterations it holds, so that
        ``render(n + 1, ...)`` continues the render."""
        _check(lib().caster_checkpoint_load(self._h, os.fspath(path).encode()))
        return self.iterations

    def primary_ids(self, it=1):
        ids = np.empty((self.height, self.width), np.int32)
        _check(lib().caster_primary_ids(self._h, it, ids.ctypes.data_as(C.POINTER(C.c_int32))))
        return ids

    def trace_rays(self, origins, dirs, tmax):
        """Closest hits (primitive ids, distances) of the caller's rays: (n, 3), (n, 3), (n,) float32."""
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        tm = np.ascontiguousarray(tmax, np.float32).reshape(-1)
        n = len(tm)
        prim = np.empty(n, np.int32)
        t = np.empty(n, np.float32)
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        _check(lib().caster_trace_rays(self._h, n, fp(o), fp(d), fp(tm), prim.ctypes.data_as(C.POINTER(C.c_int32)), fp(t)))
        return prim, t

    def trace_rays_static(self, n, origins_ptr, dirs_ptr, tmax_ptr, prim_ptr, t_ptr):
        """trace_rays on DEVICE arrays of the caller's (addresses), through the entry point that allocates nothing."""
        _check(lib().caster_trace_rays_static(self._h, int(n), *[C.c_void_p(p) for p in (origins_ptr, dirs_ptr, tmax_ptr,
                                                                                     prim_ptr, t_ptr)]))

    FUNCTIONS = {"sample_bsdf": (0, 16, 7), "eval_bsdf": (1, 16, 4), "sample_light": (2, 6, 11), "camera_ray": (3, 3, 6),
                 "hit_record": (4, 8, 15)}

    def eval_function(self, name, rows):
        """Runs device function `name` on the (n, width) float32 rows (integers as bit patterns: see include/caster.h)."""
        fn, w_in, w_out = self.FUNCTIONS[name]
        a = np.ascontiguousarray(rows, np.float32).reshape(-1, w_in)
        out = np.empty((len(a), w_out), np.float32)
        fp = lambda x: x.ctypes.data_as(C.POINTER(C.c_float))
        _check(lib().caster_eval_function(self._h, fn, len(a), fp(a), fp(out)))
        return out

    def stats(self):
        st = Stats()
        _check(lib().caster_get_stats(self._h, C.byref(st)))
        return st.as_dict()

    def walk_counts(self):
        """What the traversal kernel fetched: {closest, any} x {node records read from memory, primitive records tested}."""
        a = (C.c_uint64 * 4)()
        _check(lib().caster_get_walk_counts(self._h, a))
        return {"closest": {"nodes": a[0], "prims": a[1]}, "any": {"nodes": a[2], "prims": a[3]}}

    def set_profiling(self, enable=True):
        _check(lib().caster_set_profiling(self._h, 1 if enable else 0))

    def kernel_times(self):
        """{class: (total ms, launches)} since profiling was enabled."""
        kt = KernelTimes()
        _check(lib().caster_get_kernel_times(self._h, C.byref(kt)))
        return {n: (kt.ms[i], kt.launches[i]) for i, n in enumerate(KERNEL_CLASSES)}

    def reset_stats(self):
        _check(lib().caster_reset_stats(self._h))

    @property
    def iterations(self):
        return lib().caster_iterations(self._h)

    def close(self):
        if getattr(self, "_h", None):
            lib().caster_end(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_image(path, image):
    """Writes an (H, W, 3) float32 array, row 0 = bottom scanline, as .png / .exr / .pfm."""
    a = np.ascontiguousarray(image, dtype=np.float32)
    if a.ndim != 3 or a.shape[2] != 3:
        raise CasterError(1, "write_image: expected an (H, W, 3) array")
    _check(lib().caster_write_image(os.fspath(path).encode(), a.shape[1], a.shape[0],
                                    a.ctypes.data_as(C.POINTER(C.c_float))))


# ---- the reference's free-function boundary (pathtracer.h:8-10) ---------------
_current = None


def BeginRender(scene, width=None, height=None, epsilon=None, device=0, iters_in_flight=0, lookahead=0):
    """``BeginRender(scene, w, h, eps)``: uploads the scene.  ``width``, ``height`` and
    ``epsilon`` are accepted for signature compatibility and must match the scene's own
    (the reference always passes ``config.width/height/epsilon`` back in,
    CudaRender.cpp:37).  ``lookahead`` > 1: ``Render`` calls for consecutive iterations are
    traced that many at a time (``caster_set_lookahead``)."""
    global _current
    if width is not None and (width != scene.width or height != scene.height):
        raise CasterError(1, "BeginRender: width/height differ from the scene's")
    if epsilon is not None and abs(epsilon - scene.info.epsilon) > 0:
        raise CasterError(1, "BeginRender: epsilon differs from the scene's")
    if _current is not None:
        _current.close()
    _current = Renderer(scene, device, max(iters_in_flight, lookahead), 1 if lookahead > 1 else 0)
    if lookahead > 1 and scene.integrator not in (3, 4):   # (light tracing and BDPT add to the film while they trace)
        _current.set_lookahead(lookahead)
    return _current


def Render(scene, width, height, camera, iteration, reset, output=None):
    """``Render(scene, w, h, camera, iter, reset, output)``: one sample per pixel for
    iteration ``iteration`` (1-based), then ``output`` receives the tonemapped running
    mean, as the reference's ``Output`` kernel writes it.  ``camera`` is ignored, as in
    the reference (its upload is commented out, pathtracer.cu:2692)."""
    if _current is None:
        raise CasterError(1, "Render called before BeginRender")
    _current.render(iteration, 1, reset)
    img = _current.read_film(1)
    if output is not None:
        output[...] = img
    return img


def EndRender():
    global _current
    if _current is not None:
        _current.close()
        _current = None


class CudaRender:
    """Mirror of the reference's driver object (CudaRender.h:9-19), headless."""

    _instance = None

    @classmethod
    def getInstance(cls):
        if cls._instance is None:
            cls._instance = cls()
        return cls._instance

    def __init__(self):
        self.scene = None
        self.renderer = None
        self.iteration = 0
        self.image = None

    def InitCudaScene(self, file, device=0, iters_in_flight=1):
        self.scene = Scene(file)
        self.renderer = Renderer(self.scene, device, iters_in_flight)
        self.iteration = 0

    def renderImage(self):
        self.iteration += 1  # the reference increments before rendering (CudaRender.cpp:49)
        self.renderer.render(self.iteration, 1, False)
        self.image = self.renderer.read_film(1)
        return self.image

    def GetRenderReverseImage(self):
        return self.renderImage()[::-1].copy()

    def GetConfig(self):
        i = self.scene.info
        return {"width": i.width, "height": i.height, "epsilon": i.epsilon}

    def GetIteration(self):
        return self.iteration

    def SaveImage(self, path):
        """Writes the last image as an 8-bit PNG, top row first (imageio.cpp:62-79)."""
        write_image(path, self.image)
