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

Rank 0 prints one JSON line.  `roofline` prices the dominant kernel against
HBM bandwidth with ALGORITHMIC bytes (32 B per BVH node visited + 36 B per
triangle tested + 16/32 B per sphere/line, plus the ray/hit records the kernel
itself moves), the per-ray visit counts being those of the reference's own
traversal order, counted by the CPU oracle on a sample of the same workload.
`cpu_baseline` is that oracle (a transcription of the reference integrators;
the reference itself does not compile) timed on this host's cores.
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


def pmc_traffic_per_launch(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same
    command (profiles/r01_pmc.json: per-dispatch averages of FETCH_SIZE and WRITE_SIZE, in KiB).
    FETCH_SIZE counts half the bytes on gfx950, for 16-byte-per-lane streaming reads and for the
    128-byte record gathers of the traversal alike (tools/calib/fetch_calib.hip measures both
    against known byte counts), so it is doubled; WRITE_SIZE is exact.  None without the file."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc.json")) as f:
            k = json.load(f)[kernel]
        return (2.0 * k["FETCH_SIZE"]["avg"] + k["WRITE_SIZE"]["avg"]) * 1024.0
    except (OSError, KeyError, ValueError):
        return None


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
    ap.add_argument("--spp-per-step", type=int, default=192)
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
    spp = a.spp_per_step
    sharding = importlib.import_module("cuda-pathtracer_amd.sharding")
    r = caster.Renderer(scene, dev_index, a.iters_in_flight, a.streams)
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

    barrier()
    torch.cuda.synchronize()
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
            "kernel_launches": {k: v[1] for k, v in kt.items()},
        }
        counts = None
        if not a.no_cpu:
            import oracle_lib as ol
            o = ol.Oracle(path)
            threads = a.cpu_threads or min(os.cpu_count() or 1, 16)
            c = ol.Counts()
            # bounded sample (some 15 s of CPU work): 32 scanlines out of every 128 of the frame, 4 spp;
            # the oracle hands out whole scanlines, so a block of 32 keeps 16 threads busy
            cpu_spp = 4
            sample_rows = [(y, min(y + 32, scene.height)) for y in range(0, scene.height, 128)]
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
            cpu_model = ""
            try:
                with open("/proc/cpuinfo") as f:
                    cpu_model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "")
            except OSError:
                pass
            out["cpu_baseline"] = {
                "value": n_cpu / tc / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "port", "cpu_model": cpu_model,
                "sample": "%d samples: %d spp on 32 scanlines out of every 128 of the same %dx%d frame" % (
                    n_cpu, cpu_spp, scene.width, scene.height),
                "single_thread_msamples_per_s": c1.samples / tc1 / 1e6,
                "mrays_per_s": sum(counts["rays"]) / tc / 1e6,
            }
        # Roofline, by kernel symbol as rocprofv3 reports it: k_trace<false> runs the extension and
        # the probe rays, k_trace<true> the shadow rays, k_shade_* one path vertex per extension ray.
        # The dominant kernel is the one with the largest total time in the timed region.  With
        # several streams a launch shares the GPU with the other streams' kernels while it runs, so
        # `exclusive` repeats the measurement on one stream (same batch size), after the timed region.
        shade_name = {0: "k_shade_ao", 1: "k_shade_pt<false>", 2: "k_shade_vpt", 3: "k_shade_lt", 4: "k_bdpt"}[scene.integrator]
        if scene.integrator == 1 and kt["probe"][1] == 0:
            shade_name = "k_shade_pt<true>"   # small batches: the probe rays were among the shadow rays, no probe launches
        symbols = {"k_trace<false>": (("extend", 0), ("probe", 2)), "k_trace<true>": (("shadow", 1),),
                   shade_name: (("shade", None),)}

        def algorithmic_bytes(sym, rays_now):
            """Algorithmic bytes the launches of `sym` had to move for `rays_now` = this rank's ray
            counts [extension, shadow, probe, transmittance]."""
            if counts is None:
                return None
            if sym == shade_name:
                # per vertex: ray + hit read (32 + 16 B), path state read + write (2 x 32 B); per accepted
                # hit: shading fetch 96 B + indices 20 B + material 72 B; per emitted ray: 32 B written
                vertices = rays_now[0]
                hit_frac = counts["hits"][0] / max(counts["rays"][0], 1)
                emitted = max(rays_now[0] - total_samples_rank, 0) + rays_now[1] + rays_now[2]
                return vertices * (48.0 + 64.0) + vertices * hit_frac * 188.0 + emitted * 32.0
            record = {"extend": 32 + 16, "shadow": 32 + 4, "probe": 32 + 4 + 16}
            return sum(rays_now[kind] * (algorithmic_bytes_per_ray(counts, kind) + record[c]) for c, kind in symbols[sym])

        def roofline_of(sym, times, rays_now):
            ms = sum(times[c][0] for c, _ in symbols[sym])
            launches = sum(times[c][1] for c, _ in symbols[sym])
            alg = algorithmic_bytes(sym, rays_now)
            r = {"kernel": sym, "total_ms": ms, "launches": launches, "avg_launch_ms": ms / max(launches, 1)}
            if alg is not None and ms > 0:
                r.update(achieved=alg / (ms * 1e-3) / 1e9, frac=alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         algorithmic_bytes_per_launch=alg / max(launches, 1))
            else:
                r.update(achieved=None, frac=None)
            return r

        total_samples_rank = st["samples"]
        per_kernel = [roofline_of(sym, kt, st["rays"]) for sym in symbols]
        dom = max(per_kernel, key=lambda r: r["total_ms"])
        roof = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": pmc_traffic_per_launch(dom["kernel"])}
        roof.update(dom)
        if counts is not None:
            kinds = [kind for _, kind in symbols[dom["kernel"]] if kind is not None] or [0]
            o_rays = sum(counts["rays"][kind] for kind in kinds)
            roof.update(nodes_per_ray=sum(counts["nodes"][kind] for kind in kinds) / max(o_rays, 1),
                        tris_per_ray=sum(counts["tris"][kind] for kind in kinds) / max(o_rays, 1))
        # one-stream repeat: each launch has the GPU to itself
        try:
            if a.no_exclusive:
                raise caster.CasterError(0, "skipped (--no-exclusive)")
            r1 = caster.Renderer(scene, dev_index, r.iters_in_flight, 1)
            if rows != (0, scene.height):
                r1.set_row_stripes(*rows)
            n1 = r.iters_in_flight
            r1.render(1, n1, True)
            r1.reset_stats()
            r1.set_profiling(True)
            t1 = time.perf_counter()
            r1.render(1 + n1, 2 * n1, False)
            dt1 = time.perf_counter() - t1
            st1, kt1 = r1.stats(), r1.kernel_times()
            r1.close()
            total_samples_rank = st1["samples"]
            ex = [roofline_of(sym, kt1, st1["rays"]) for sym in symbols]
            roof["exclusive"] = {"streams": 1, "msamples_per_s": st1["samples"] / dt1 / 1e6,
                                 "kernels": [{k: v for k, v in e.items() if k in ("kernel", "avg_launch_ms", "achieved", "frac", "launches")}
                                             for e in ex]}
        except caster.CasterError as e:
            roof["exclusive"] = {"error": str(e)}
        roof["kernels"] = [{k: v for k, v in e.items() if k in ("kernel", "total_ms", "avg_launch_ms", "achieved", "frac", "launches")}
                           for e in per_kernel]
        out["roofline"] = roof
        print(json.dumps(out))
    r.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
