#!/usr/bin/env python3
"""Headline benchmark: Msamples/s (and Mrays/s) of path tracing + MIS on the
bmw-m6 stand-in scene at 1920x1080, at 1/2/4/8 MI355X.

  python bench.py --gpus N --steps K --warmup W

One "step" is one batch of `--spp-per-step` samples per pixel over the whole
frame through the wavefront renderer (generate -> [extend -> shade ->
shadow/probe]* -> finish), inputs resident in HBM.  With N > 1 (launched by
torch.distributed.run, one rank per GPU) every rank renders its own, disjoint
iterations of the full frame (weak scaling: per-GPU work fixed) and the films
are summed onto rank 0 with one RCCL reduce inside the timed region.

Started without a launcher, `--gpus N` starts its own N ranks (torch.distributed.run as a child process,
before anything here touches a GPU) and passes rank 0's line on.

Rank 0 prints one JSON line.  `roofline` holds the kernel that takes most of the GPU's time against the roof
that binds it: for the traversal kernel that is vector-instruction issue (instructions per launch from the
committed PMC passes over the launch duration measured in this run, against one wave64 instruction per SIMD
per four clocks; the clock is sampled in this run).  Beside it, as `bytes`, the bytes the kernel itself requests
(GPU-side counters: 112 B per 4-wide node record fetched + 48 B per primitive record tested + the ray and hit
records moved) over the same duration against L2 bandwidth, the L2-side and HBM-side bytes of the PMC passes,
and what the reference's own traversal order would have requested (`work_avoided`; visit counts of the CPU
oracle on a sample of the same workload).  Every number that comes from the committed PMC file rather than
from this run says so (`source`).
`cpu_baseline` is that oracle (a transcription of the reference integrators; the reference itself does not
compile) timed on this host's cores.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak
L2_PEAK_GBS = 34500.0   # aggregate L2 bandwidth of the eight XCDs
# what each synthetic scene of tools/gen_scenes.py stands in for among the benchmark configurations
STAND_IN = {"car_pt": "bmw-m6 stand-in", "stairs_pt": "staircase2 stand-in", "knot_pt": "teapot-full stand-in",
            "cornell_ao": "Cornell box", "cornell_pt": "Cornell box", "cornell_vpt": "Cornell box with a homogeneous medium",
            "cornell_smoke": "Cornell box with a heterogeneous medium", "cornell_meshlight": "Cornell box with a mesh light"}


def scene_path(name, detail, width, height, depth, rank, barrier, integrator=None):
    """Generates the synthetic scene once per host (rank 0) and returns its path.  `integrator`
    renders the same scene with another of the reference's integrators (the "integrator" key)."""
    import gen_scenes
    tag = "%s_d%g_%dx%d_m%d" % (name, detail, width, height, depth)
    base = os.path.join(os.environ.get("CASTER_SCENE_CACHE", "/tmp/caster_scenes_%d" % os.getuid()), tag)
    path = os.path.join(base, "scene.json")
    if rank == 0 and not os.path.exists(path):
        gen_scenes.generate(base + ".tmp%d" % os.getpid(), name, detail, width, height, depth)
        try:
            os.rename(base + ".tmp%d" % os.getpid(), base)
        except OSError:
            pass  # another process won the race
    if integrator:
        other = os.path.join(base, "scene_%s.json" % integrator)
        if rank == 0 and not os.path.exists(other):
            with open(path) as f:
                doc = json.load(f)
            doc["integrator"] = integrator
            with open(other + ".tmp%d" % os.getpid(), "w") as f:
                json.dump(doc, f)
            os.replace(other + ".tmp%d" % os.getpid(), other)
        path = other
    barrier()
    return path


def algorithmic_bytes_per_ray(counts, kind):
    """Algorithmic per-ray traversal bytes (32 B per node visited, 36 B per triangle, 16 B per sphere, 32 B per line tested) from the oracle's visit counts."""
    rays = counts["rays"][kind]
    if rays == 0:
        return 0.0
    return (32.0 * counts["nodes"][kind] + 36.0 * counts["tris"][kind] + 16.0 * counts["spheres"][kind]
            + 32.0 * counts["lines"][kind]) / rays


PMC_FILE = os.path.join("profiles", "r03_pmc.json")
PMC_SOURCE = ("committed PMC pass: %s, per-dispatch averages of rocprofv3 --pmc runs of this same command on the builder's "
              "own lease (tools/gpu_prof.sh); not measured in this run" % PMC_FILE)
TCP_REQ_BYTES = 128.0   # bytes per L1 -> L2 read request, calibrated with known byte counts (profiles/r03_calib_tcp.txt)
N_SIMD = 1024           # 256 CUs x 4 SIMDs
PEAK_CLOCK_HZ = 2.4e9
VALU_PEAK_PER_S = N_SIMD * PEAK_CLOCK_HZ / 4.0   # wave64 vector instructions per second: one per SIMD per four clocks


def load_pmc():
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def pmc_of(pmc, kernel):
    """What the committed PMC passes say about `kernel`, per launch (per-dispatch averages), every number labelled
    with its source.  hbm_bytes: FETCH_SIZE counts half the bytes on gfx950, for 16-byte-per-lane streaming reads and
    for the 128-byte record gathers of the traversal alike (tools/calib/fetch_calib.hip measures both against known
    byte counts), so it is doubled; WRITE_SIZE is exact; both are in KiB.  l2_request_bytes: L1 -> L2 read requests x
    128 B (calibrated the same way).  valu_issue_busy: SQ_ACTIVE_INST_VALU over the wavefronts' lifetime per SIMD (the
    SQ counts in units of four clocks, one wave64 instruction each).  None without the file or the kernel."""
    try:
        k = pmc[kernel]
        waves_per_simd = k["SQ_WAVES"]["avg"] / N_SIMD
        return {"source": PMC_SOURCE,
                "valu_wave_instructions": k["SQ_INSTS_VALU"]["avg"],
                "valu_issue_busy": k["SQ_ACTIVE_INST_VALU"]["avg"] / (k["SQ_WAVE_CYCLES"]["avg"] / waves_per_simd),
                "useful_lane_frac": k["SQ_THREAD_CYCLES_VALU"]["avg"] / (64.0 * k["SQ_ACTIVE_INST_VALU"]["avg"]),
                "l2_hit_rate": k["TCC_HIT_sum"]["avg"] / (k["TCC_HIT_sum"]["avg"] + k["TCC_MISS_sum"]["avg"]),
                "waves_per_simd": waves_per_simd,
                "hbm_bytes": (2.0 * k["FETCH_SIZE"]["avg"] + k["WRITE_SIZE"]["avg"]) * 1024.0,
                "l2_request_bytes": k["TCP_TCC_READ_REQ_sum"]["avg"] * TCP_REQ_BYTES}
    except (TypeError, KeyError, ValueError, ZeroDivisionError):
        return None


class ClockSampler:
    """Samples the shader clock the driver reports for this device (the starred level of pp_dpm_sclk) every 20 ms
    while the timed region runs: the chip does not hold its 2.4 GHz peak under this load."""

    def __init__(self, index):
        import glob
        import threading
        cards = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
        self.path = cards[index] if index < len(cards) else None
        self.samples = []
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _read(self):
        try:
            with open(self.path) as f:
                for line in f:
                    if "*" in line:
                        return float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
        except (OSError, ValueError, IndexError):
            pass
        return None

    def _run(self):
        while not self._stop.is_set():
            v = self._read()
            if v is not None:
                self.samples.append(v)
            self._stop.wait(0.02)

    def start(self):
        if self.path:
            self._thread.start()

    def stop(self):
        self._stop.set()
        if self.path:
            self._thread.join()
        if not self.samples:
            return None
        xs = sorted(self.samples)
        return {"mean": sum(xs) / len(xs), "min": xs[0], "median": xs[len(xs) // 2], "max": xs[-1], "samples": len(xs),
                "source": "measured in this run: %s sampled every 20 ms over the timed region" % self.path}


def launch_ranks(n, argv):
    """Runs this script under torch.distributed.run with `n` ranks on this node (one per GPU) as a
    child process and returns its exit status; the ranks' output passes through unchanged."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default="car_pt")
    ap.add_argument("--detail", type=float, default=1.0)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--integrator", default=None, help="render the scene with this integrator instead of its own (experiments)")
    ap.add_argument("--spp-per-step", type=int, default=0,
                    help="samples per pixel per step (0: one batch on every stream, i.e. streams x iterations in flight)")
    ap.add_argument("--iters-in-flight", type=int, default=0, help="iterations per batch of paths (0: library default)")
    ap.add_argument("--streams", type=int, default=0, help="HIP streams batches are dealt to (0: library default)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline (roofline then uses no counts)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--no-exclusive", action="store_true", help="skip the one-stream repeat (profiling runs)")
    ap.add_argument("--shard", choices=["spp", "tile"], default="spp")
    ap.add_argument("--backend", default="nccl", help="process-group backend (gloo + --share-gpu rehearses N>1 on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="every rank uses device 0 (rehearsal only)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Started without a launcher: start one rank per GPU ourselves -- before anything here has
        # touched the GPU -- and pass on rank 0's line and the launcher's exit status.
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d: start it with --nproc-per-node %d, or without a "
                         "launcher (it then starts its own ranks)" % (a.gpus, world, a.gpus))
    import torch
    dist = None
    dev_index = 0 if a.share_gpu else local_rank
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(dev_index)
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(a.backend)
    coll_dev = "cuda:%d" % dev_index if a.backend == "nccl" else "cpu"

    def barrier():
        if dist is not None:
            dist.barrier()

    caster = importlib.import_module("cuda-pathtracer_amd")
    path = scene_path(a.scene, a.detail, a.width, a.height, a.depth, rank, barrier, a.integrator)
    scene = caster.Scene(path)
    sharding = importlib.import_module("cuda-pathtracer_amd.sharding")
    # The library's default pools are sized for a host that shares the card (64 M paths per stream, a quarter of the
    # free memory at most); a benchmark rank has its GPU to itself and asks for the 128 M paths per stream that are
    # the last percent faster -- unless the ranks share one card (rehearsals).
    iters = a.iters_in_flight
    if iters <= 0 and not a.share_gpu:
        slots = ((scene.width + 7) // 8) * ((scene.height + 7) // 8) * 64
        iters = max(1, min(64, (128 << 20) // slots))
    r = caster.Renderer(scene, dev_index, iters, a.streams)
    r_streams = a.streams or 3
    spp = a.spp_per_step or r_streams * r.iters_in_flight
    npix = scene.width * scene.height
    rows = (0, scene.height)
    if world > 1 and a.shard == "tile":
        rows = sharding.row_stripes(scene.height, rank, world)
        r.set_row_stripes(*rows)

    film = torch.empty(npix * 3, dtype=torch.float32, device="cuda:%d" % dev_index) if world > 1 else None

    def first_iter(step):
        # spp: block `step * world + rank` of the iteration sequence (sharding.spp_blocks with
        # block = spp); tile: every rank renders the same iterations for its own rows
        if a.shard == "tile":
            return 1 + step * spp
        return 1 + (step * world + rank) * spp

    step_no = 0
    for _ in range(a.warmup):
        r.render_async(first_iter(step_no), spp, False)
        step_no += 1
    r.sync()
    r.reset_stats()
    r.render(1, 0, True)          # clear the film and the iteration count
    r.set_profiling(True)

    clock = ClockSampler(dev_index)
    barrier()
    torch.cuda.synchronize()
    clock.start()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        r.render_async(first_iter(step_no), spp, False)
        step_no += 1
    if world > 1:
        r.read_film_device(film.data_ptr(), 2)
        if coll_dev == "cpu":
            host_film = film.cpu()
            sharding.reduce_film(host_film, dist, 0)
        else:
            sharding.reduce_film(film, dist, 0)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    sclk = clock.stop()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    st = r.stats()
    kt = r.kernel_times()
    rays = torch.tensor(st["rays"] + [st["samples"], st["invalid"], st["stack_overflow"]], dtype=torch.float64)
    if world > 1:
        rays = rays.to(coll_dev)
        dist.all_reduce(rays)
        rays = rays.cpu()
    rays = [float(x) for x in rays]
    total_samples = rays[4]
    total_rays = sum(rays[:4])
    # what the process group really is: backend, size, and every rank's device, gathered onto rank 0
    group = None
    if world > 1:
        mine = {"rank": rank, "device_index": dev_index, "device": torch.cuda.get_device_name(dev_index),
                "samples": st["samples"], "seconds": time.perf_counter() - t0}
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
        group = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks": ranks,
                 "film_reduce": "torch.distributed.reduce(sum) of %d floats onto rank 0, inside the timed region" % (npix * 3)}

    if rank == 0:
        out = {
            "metric": "Msamples/s %s %s %s" % (
                STAND_IN.get(a.scene, a.scene).replace(" stand-in", ""),
                "1080p" if (scene.width, scene.height) == (1920, 1080) else "%dx%d" % (scene.width, scene.height),
                {1: "path+MIS"}.get(scene.integrator, caster.INTEGRATOR_NAMES[scene.integrator])),
            "value": total_samples / dt / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak" if a.shard == "spp" else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "mrays_per_s": total_rays / dt / 1e6,
            "rays": {"extension": rays[0], "shadow": rays[1], "probe": rays[2], "transmittance": rays[3]},
            "invalid_samples": rays[5],
            "stack_overflow": rays[6],
            "config": {
                "workload": "%s (%s, %d primitives, %d BVH nodes) %dx%d %s maxDepth %d" % (
                    a.scene, STAND_IN.get(a.scene, "synthetic scene"), scene.info.n_prims, scene.info.n_nodes,
                    scene.width, scene.height,
                    caster.INTEGRATOR_NAMES[scene.integrator], scene.info.max_depth),
                "spp_per_step": spp,
                "sharding": ("none" if world == 1 else
                             "spp (disjoint iterations per rank, one film reduce)" if a.shard == "spp" else
                             "tile (slabs of 8 scanlines dealt round-robin over the ranks, one film reduce)"),
            },
            "kernel_ms": {k: v[0] for k, v in kt.items()},
            "kernel_ms_note": "OVERLAPPED: launch durations summed over %d streams that share the GPU, so the classes add up to more "
                              "than steps x ms_per_step; durations of launches that have the GPU to themselves are under "
                              "roofline.exclusive" % r_streams,
            "kernel_launches": {k: v[1] for k, v in kt.items()},
            "sclk_mhz": sclk,
        }
        if group is not None:
            out["process_group"] = group
        counts = None
        if not a.no_cpu and world == 1:   # (the CPU baseline belongs to the one-GPU line)
            import oracle_lib as ol
            o = ol.Oracle(path)
            # the oracle on this host's cores: as many threads as there are (64 at most), and one
            host_cpus = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            threads = a.cpu_threads or min(host_cpus, 64)
            c = ol.Counts()
            # bounded sample (some 15 s of CPU work): `band` scanlines out of every 128 of the frame, 4 spp;
            # the oracle hands out whole scanlines, so a band holds one scanline per thread at least
            cpu_spp = 4
            band = max(32, min(threads, 128))
            sample_rows = [(y, min(y + band, scene.height)) for y in range(0, scene.height, max(128, band))]
            t0 = time.perf_counter()
            for y0, y1 in sample_rows:
                o.render(1, cpu_spp, threads, counts=c, rows=(y0, y1))
            tc = time.perf_counter() - t0
            n_cpu = c.samples
            c1 = ol.Counts()
            y_mid = scene.height // 2
            t0 = time.perf_counter()
            o.render(1, 1, 1, counts=c1, rows=(y_mid, min(y_mid + 16, scene.height)))
            tc1 = time.perf_counter() - t0
            counts = c.as_dict()
            cpu_model, sockets, cpu_id = "", set(), {}
            try:
                with open("/proc/cpuinfo") as f:
                    for l in f:
                        key, _, val = l.partition(":")
                        key, val = key.strip(), val.strip()
                        if key == "model name" and not cpu_model:
                            cpu_model = val
                        if key in ("cpu family", "model", "stepping", "cpu MHz", "cache size") and key not in cpu_id:
                            cpu_id[key] = val
                        if key == "physical id":
                            sockets.add(val)
            except OSError:
                pass
            if cpu_id:   # (a virtual machine may give no more than the vendor's family name as the model string)
                cpu_model = "%s (family %s model %s stepping %s, %s MHz, cache %s)" % (
                    cpu_model, cpu_id.get("cpu family", "?"), cpu_id.get("model", "?"), cpu_id.get("stepping", "?"),
                    cpu_id.get("cpu MHz", "?"), cpu_id.get("cache size", "?"))
            out["cpu_baseline"] = {
                "value": n_cpu / tc / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "port", "cpu_model": cpu_model,
                "host_logical_cpus": os.cpu_count(), "cpus_this_process_may_use": host_cpus, "sockets": len(sockets) or None,
                "sample": "%d samples: %d spp on %d scanlines out of every %d of the same %dx%d frame" % (
                    n_cpu, cpu_spp, band, max(128, band), scene.width, scene.height),
                "single_thread_msamples_per_s": c1.samples / tc1 / 1e6,
                "mrays_per_s": sum(counts["rays"]) / tc / 1e6,
            }
        # Roofline, by kernel symbol as rocprofv3 reports it: k_trace<false> runs the extension and the probe rays,
        # k_trace<true> the shadow rays, k_shade_* one path vertex per extension ray that hit something.
        #   requested bytes  what the kernel itself asks the memory system for, from counters it keeps on the GPU:
        #                    112 B per 4-wide node record fetched (the top of the tree is served from LDS and not
        #                    counted) + 48 B per primitive record tested + the ray and hit records it moves;
        #   duration         of a launch that has the GPU to itself (`exclusive`: the one-stream repeat after the
        #                    timed region; with three streams a launch's wall time includes the time it shares the
        #                    CUs with the other streams' kernels -- `overlapped`);
        #   roof             the traversal gathers 128-byte records, most of them from L2 and the per-CU caches (66 MB
        #                    of nodes and primitives against 32 MB of L2 and 256 MB of Infinity Cache): L2 bandwidth,
        #                    34.5 TB/s; the shade stage streams path state and gathers 128-byte shading records from
        #                    a 176 MB table: HBM, 8 TB/s;
        #   traffic          HBM-side bytes per launch from the committed PMC passes (fabric reads and writes), and
        #                    traffic / duration against the HBM peak (`hbm_frac`);
        #   work_avoided     the bytes the REFERENCE's own traversal order would have requested for the same rays
        #                    (32 B per binary node visited + 36 B per triangle tested, visit counts of the oracle on a
        #                    sample of this workload) over the requested bytes: what the 4-wide, nearest-first,
        #                    entry-distance-culled walk saves.
        shade_name = {0: "k_shade_ao", 1: "k_shade_pt<false>", 2: "k_shade_vpt", 3: "k_shade_lt", 4: "k_bdpt"}[scene.integrator]
        if scene.integrator == 1 and kt["probe"][1] == 0:
            shade_name = "k_shade_pt<true>"   # small batches: the probe rays were among the shadow rays, no probe launches
        symbols = {"k_trace<false>": (("extend", 0), ("probe", 2)), "k_trace<true>": (("shadow", 1),),
                   shade_name: (("shade", None),)}
        ROOF = {"k_trace<false>": ("l2", L2_PEAK_GBS), "k_trace<true>": ("l2", L2_PEAK_GBS), shade_name: ("hbm", HBM_PEAK_GBS)}

        def requested_bytes(sym, rays_now, walk_now, samples_now):
            """Bytes the launches of `sym` asked for, for this renderer's ray counts [extension, shadow, probe,
            transmittance] and traversal counters."""
            if sym == shade_name:
                # per queue entry: the hit record is read to see whether a vertex is needed (16 B); per vertex:
                # ray + hit + slot (52 B), path state read and written (2 x 32 B), the pending terms written and
                # read back (96 B), the shading record (128 B) and the material (96 B); per emitted ray: written to
                # the staging region, read back and written to the queue (3 x 36 B)
                entries = rays_now[0]
                hit_frac = (counts["hits"][0] / max(counts["rays"][0], 1)) if counts else 0.5
                vertices = entries * hit_frac
                emitted = max(rays_now[0] - samples_now, 0) + rays_now[1] + rays_now[2]
                return entries * 16.0 + vertices * (52.0 + 64.0 + 96.0 + 128.0 + 96.0) + emitted * 108.0
            if sym == "k_trace<true>":
                return walk_now["any"]["nodes"] * 112.0 + walk_now["any"]["prims"] * 48.0 + rays_now[1] * (32.0 + 4.0)
            # extension rays write a 16-byte hit record, probe rays too
            return (walk_now["closest"]["nodes"] * 112.0 + walk_now["closest"]["prims"] * 48.0
                    + (rays_now[0] + rays_now[2] + rays_now[3]) * (32.0 + 16.0))

        def reference_walk_bytes(sym, rays_now):
            if counts is None or sym == shade_name:
                return None
            record = {"extend": 32 + 16, "shadow": 32 + 4, "probe": 32 + 4 + 16}
            return sum(rays_now[kind] * (algorithmic_bytes_per_ray(counts, kind) + record[c]) for c, kind in symbols[sym])

        pmc = load_pmc() if a.scene == "car_pt" and (scene.width, scene.height) == (1920, 1080) else None   # (the profiled command)
        sclk_hz = sclk["mean"] * 1e6 if sclk else None

        def roofline_of(sym, times, rays_now, walk_now, samples_now):
            """One kernel against its roofs.  Measured in this run: launch durations (HIP events on the renderer's own
            streams), the bytes the kernel requests (its own counters), the clock.  From the committed PMC passes, and
            labelled so (`pmc.source`): vector instructions per launch, lane utilisation, HBM-side and L2-side bytes."""
            ms = sum(times[c][0] for c, _ in symbols[sym])
            launches = sum(times[c][1] for c, _ in symbols[sym])
            req = requested_bytes(sym, rays_now, walk_now, samples_now)
            level, peak = ROOF[sym]
            per_launch_s = ms * 1e-3 / max(launches, 1)
            r = {"kernel": sym, "total_ms": ms, "launches": launches, "avg_launch_ms": ms / max(launches, 1),
                 "requested_bytes_per_launch": req / max(launches, 1)}
            by = {"bound": level, "peak": peak, "unit": "GB/s",
                  "achieved": req / (ms * 1e-3) / 1e9 if ms > 0 and req > 0 else None,
                  "what": "bytes the kernel's loads and stores request (its own counters, this run) over the launch duration "
                          "(this run), against the memory level that serves most of them; the per-CU cache serves most of "
                          "the traversal's requests, so this prices L1-served bytes against the L2 roof -- see l2_side"}
            by["frac"] = by["achieved"] / peak if by["achieved"] is not None else None
            ref = reference_walk_bytes(sym, rays_now)
            by["work_avoided"] = ref / req if ref and req else None
            k = pmc_of(pmc, sym)
            if k is not None and per_launch_s > 0:
                # (the PMC passes ran the same command: launches of the same average size)
                k["valu_issue"] = {"achieved": k["valu_wave_instructions"] / per_launch_s / 1e9, "peak": VALU_PEAK_PER_S / 1e9,
                                   "unit": "G wave64 vector instructions/s",
                                   "frac": k["valu_wave_instructions"] / per_launch_s / VALU_PEAK_PER_S,
                                   "frac_at_measured_clock": (k["valu_wave_instructions"] / per_launch_s / (N_SIMD * sclk_hz / 4.0)
                                                              if sclk_hz else None),
                                   "what": "instructions per launch (PMC pass) over this run's launch duration, against one "
                                           "wave64 instruction per SIMD per four clocks at the 2.4 GHz peak clock / at the "
                                           "clock sampled in this run"}
                k["hbm_side"] = {"bytes_per_launch": k["hbm_bytes"], "over_requested": k["hbm_bytes"] / r["requested_bytes_per_launch"] if req > 0 else None,
                                 "achieved": k["hbm_bytes"] / per_launch_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": k["hbm_bytes"] / per_launch_s / 1e9 / HBM_PEAK_GBS}
                k["l2_side"] = {"bytes_per_launch": k["l2_request_bytes"], "achieved": k["l2_request_bytes"] / per_launch_s / 1e9,
                                "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": k["l2_request_bytes"] / per_launch_s / 1e9 / L2_PEAK_GBS,
                                "what": "L1 -> L2 read requests x 128 B (calibrated, profiles/r03_calib_tcp.txt) over this run's launch duration"}
            r["bytes"] = by
            r["pmc"] = k
            return r

        def top_level(dom):
            """The `roofline` object of the line: the dominant kernel against the roof that binds it.  For the traversal
            kernel that is vector-instruction issue (its SIMDs' vector ALUs issue for most of the launch while its
            records come from the caches); without the PMC file, or for a streaming kernel, the byte roof."""
            k = dom.get("pmc")
            t = {"kernel": dom["kernel"], "avg_launch_ms": dom["avg_launch_ms"], "launches": dom["launches"],
                 "requested_bytes_per_launch": dom["requested_bytes_per_launch"]}
            if k is not None and dom["kernel"].startswith("k_trace"):
                t.update({"bound": "valu_issue", "achieved": k["valu_issue"]["achieved"], "peak": k["valu_issue"]["peak"],
                          "unit": k["valu_issue"]["unit"], "frac": k["valu_issue"]["frac"],
                          "frac_at_measured_clock": k["valu_issue"]["frac_at_measured_clock"],
                          "useful_lane_frac": k["useful_lane_frac"], "traffic": k["hbm_bytes"],
                          "measured_in_this_run": ["avg_launch_ms", "launches", "requested_bytes_per_launch", "sclk_mhz", "bytes.achieved"],
                          "from_committed_pmc_pass": {"source": PMC_SOURCE,
                                                      "fields": ["achieved (instructions per launch)", "frac", "useful_lane_frac", "traffic",
                                                                 "bytes.hbm_side", "bytes.l2_side"]},
                          "note": "the kernel's SIMDs issue vector instructions for frac of the launch at the peak clock "
                                  "(frac_at_measured_clock at the clock the chip held), with useful_lane_frac of each "
                                  "instruction's lanes active; its bytes are secondary (`bytes`)"})
                t["bytes"] = dict(dom["bytes"], hbm_side=k["hbm_side"], l2_side=k["l2_side"])
            else:
                t.update({"bound": dom["bytes"]["bound"], "achieved": dom["bytes"]["achieved"], "peak": dom["bytes"]["peak"],
                          "unit": "GB/s", "frac": dom["bytes"]["frac"], "traffic": k["hbm_bytes"] if k else None,
                          "measured_in_this_run": ["avg_launch_ms", "launches", "requested_bytes_per_launch", "achieved", "frac"],
                          "from_committed_pmc_pass": {"source": PMC_SOURCE, "fields": ["traffic"]} if k else None})
            return t

        walk = r.walk_counts()
        iters_in_flight = r.iters_in_flight
        r.close()    # (its path pools take most of the card: the one-stream repeat below needs a third as much again)
        per_kernel = [roofline_of(sym, kt, st["rays"], walk, st["samples"]) for sym in symbols]
        # every launch of the timed region against the wall time: what the three kernels together asked for per second
        pipeline_req = sum(k["requested_bytes_per_launch"] * k["launches"] for k in per_kernel)
        have_pmc = pmc is not None and world == 1 and all(k.get("pmc") for k in per_kernel)
        pipeline = {"requested_bytes_per_step": pipeline_req / max(a.steps, 1), "requested_rate_gbs": pipeline_req / dt / 1e9}
        if have_pmc:
            insts = sum(k["pmc"]["valu_wave_instructions"] * k["launches"] for k in per_kernel)
            for cat, sym in (("generate", "k_generate"), ("finish", "k_finish")):
                if pmc_of(pmc, sym):
                    insts += pmc_of(pmc, sym)["valu_wave_instructions"] * out["kernel_launches"].get(cat, 0)
            hbm = sum(k["pmc"]["hbm_bytes"] * k["launches"] for k in per_kernel)
            pipeline.update({
                "source": PMC_SOURCE + "; multiplied by this run's launch counts and divided by this run's wall time",
                "valu_issue": {"wave_instructions_per_step": insts / max(a.steps, 1), "achieved": insts / dt / 1e9,
                               "peak": VALU_PEAK_PER_S / 1e9, "unit": "G wave64 vector instructions/s", "frac": insts / dt / VALU_PEAK_PER_S,
                               "frac_at_measured_clock": insts / dt / (N_SIMD * sclk_hz / 4.0) if sclk_hz else None},
                "hbm_traffic_rate_gbs": hbm / dt / 1e9, "hbm_frac": hbm / dt / 1e9 / HBM_PEAK_GBS})
        roof = {"pipeline": pipeline,
                "overlapped": {"streams": r_streams, "kernels": per_kernel,
                               "note": "launch durations of the timed region: several streams share the GPU, a launch's duration "
                                       "includes the time it waits for the other streams' kernels"}}
        # one-stream repeat: each launch has the GPU to itself
        try:
            if a.no_exclusive or world > 1:
                raise caster.CasterError(0, "skipped (--no-exclusive, or several ranks: the other ranks would wait)")
            r1 = caster.Renderer(scene, dev_index, iters_in_flight, 1)
            if rows != (0, scene.height):
                r1.set_row_stripes(*rows)
            n1 = iters_in_flight
            r1.render(1, n1, True)
            r1.reset_stats()
            r1.set_profiling(True)
            t1 = time.perf_counter()
            r1.render(1 + n1, 2 * n1, False)
            dt1 = time.perf_counter() - t1
            st1, kt1, walk1 = r1.stats(), r1.kernel_times(), r1.walk_counts()
            r1.close()
            ex = [roofline_of(sym, kt1, st1["rays"], walk1, st1["samples"]) for sym in symbols]
            # the top level: the kernel that takes most of the GPU's time when every launch has the GPU to itself
            # (in the timed region the streams' kernels share the CUs, and whose launches appear longest there
            # changes from run to run)
            dom = max(ex, key=lambda k: k["total_ms"])
            roof.update(top_level(dom))
            roof["exclusive"] = {"streams": 1, "msamples_per_s": st1["samples"] / dt1 / 1e6, "kernels": ex,
                                 "requested_rate_gbs": sum(k["requested_bytes_per_launch"] * k["launches"] for k in ex) / dt1 / 1e9}
            roof["duration"] = "one-stream repeat after the timed region (a launch has the GPU to itself)"
        except caster.CasterError as e:
            dom = max(per_kernel, key=lambda k: k["total_ms"])
            roof.update(top_level(dom))
            roof["exclusive"] = {"error": str(e)}
            roof["duration"] = "timed region (launches of several streams overlap)"
        if counts is not None:
            o_rays = sum(counts["rays"][:3])
            roof["reference_walk"] = {"nodes_per_ray": sum(counts["nodes"][:3]) / max(o_rays, 1),
                                      "tris_per_ray": sum(counts["tris"][:3]) / max(o_rays, 1)}
        all_rays = max(st["rays"][0] + st["rays"][1] + st["rays"][2], 1)
        roof["gpu_walk"] = {"node_fetches_per_ray": (walk["closest"]["nodes"] + walk["any"]["nodes"]) / all_rays,
                            "prim_tests_per_ray": (walk["closest"]["prims"] + walk["any"]["prims"]) / all_rays}
        out["roofline"] = roof
        print(json.dumps(out))
    r.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
