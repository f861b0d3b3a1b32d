#!/usr/bin/env python3
"""Headless render of a scene file in the reference's JSON schema to an image.

  python tools/render.py scene.json --spp 256 --out image.png
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
      tools/render.py scene.json --spp 4096 --shard spp --out image.exr

One process per GPU.  With several ranks the iterations (``--shard spp``) or bands of scanlines
(``--shard tile``) are split over the ranks (cuda-pathtracer_amd/sharding.py) and the films meet on
rank 0 in one RCCL reduce, which then writes the image: .png is tonemapped as the reference's
Output() + SavePng do, .exr and .pfm hold the linear mean.

``--checkpoint FILE`` keeps the raw film sum and the iteration count in FILE (rewritten whenever the image is),
``--resume FILE`` continues from such a file: iterations n + 1 .. n + spp are rendered on top of the n it holds, and
the result equals n + spp iterations rendered in one go (a sample depends on its pixel and iteration alone, and the
film is a sum).

``--preview-every K`` rewrites the image after every K samples per pixel (the reference's window
shows its running mean after every iteration, CudaRender.cpp:48-60): the ranks' films are reduced
at each of those points, rank 0 keeps the running sum and the other ranks start their film afresh.
"""
import argparse
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scene")
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--out", default="render.png")
    ap.add_argument("--shard", choices=["spp", "tile"], default="spp")
    ap.add_argument("--gpu-bvh", action="store_true", help="build the BVH on the GPU (same tree as the host build)")
    ap.add_argument("--iters-in-flight", type=int, default=0)
    ap.add_argument("--preview-every", type=int, default=0, metavar="K", help="rewrite the image after every K samples per pixel")
    ap.add_argument("--checkpoint", default=None, metavar="FILE", help="write the raw film sum + iteration count here whenever the image is written")
    ap.add_argument("--resume", default=None, metavar="FILE", help="continue from a checkpoint: --spp more iterations on top of those it holds")
    ap.add_argument("--backend", default="nccl", help="process-group backend (gloo + --share-gpu rehearses N > 1 on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="every rank uses device 0 (rehearsal only)")
    a = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dev = 0 if a.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    caster = importlib.import_module("cuda-pathtracer_amd")
    sharding = importlib.import_module("cuda-pathtracer_amd.sharding")
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(dev)
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(a.backend)
    scene = caster.Scene(a.scene, build_device=dev if a.gpu_bvh else None)
    r = caster.Renderer(scene, dev, a.iters_in_flight)
    base = 0              # iterations rendered before this run (their sum sits in rank 0's film)
    if a.resume:
        import struct
        with open(a.resume, "rb") as f:
            magic, version, cw, ch, base, cinteg = struct.unpack("<8sIIIIi", f.read(28))
        if rank == 0:
            r.load_checkpoint(a.resume)   # (checks magic, size and integrator)
    t0 = time.perf_counter()
    tonemapped = a.out.lower().endswith(".png")
    if world > 1:
        import torch
        film = torch.zeros(scene.width * scene.height * 3, dtype=torch.float32, device="cuda:%d" % dev)
    if world > 1 and a.shard == "tile":
        # slabs of 8 scanlines dealt round-robin, so that every rank gets the same mix of the image
        # (contiguous bands leave the rank with the dearest band to finish last)
        r.set_row_stripes(*sharding.row_stripes(scene.height, rank, world))
    # samples per pixel between two images: all of them unless previews are asked for
    step = a.spp if a.preview_every <= 0 else min(a.preview_every, a.spp)
    done = 0
    fresh = not (a.resume and rank == 0)   # this rank's film holds nothing yet
    while done < a.spp:
        n_round = min(step, a.spp - done)
        if world > 1 and a.shard == "tile":
            blocks = [(1 + base + done, n_round)]
        else:
            block = max(1, min(64, n_round // max(world, 1)))
            blocks = sharding.spp_blocks(1 + base + done, n_round, block, rank, world)
        for it, n in blocks:
            r.render_async(it, n, fresh)
            fresh = False
        r.sync()
        done += n_round
        if world == 1:
            r.save_image(a.out)
            if a.checkpoint:
                r.save_checkpoint(a.checkpoint)
        else:
            # the library works on its own streams, torch on another: meet on the host in between
            if fresh:
                film.zero_()
                torch.cuda.synchronize(dev)
            else:
                r.read_film_device(film.data_ptr(), 2)
            if a.backend == "nccl":
                sharding.reduce_film(film, dist, 0)
            else:  # films meet on the host
                host = film.cpu()
                sharding.reduce_film(host, dist, 0)
                film.copy_(host)
            torch.cuda.synchronize(dev)
            if rank == 0:
                # the summed film (rank 0's own included) back into the renderer, which divides,
                # tonemaps and writes it, and keeps adding to it in the next round
                r.render(1, 0, True)
                r.film_add_device(film.data_ptr(), base + done)
                r.save_image(a.out)
                if a.checkpoint:
                    r.save_checkpoint(a.checkpoint)
            else:
                fresh = True   # handed over: the next round starts from an empty film
        if rank == 0 and done < a.spp:
            print("preview: %d of %d spp after %.2f s -> %s" % (done, a.spp, time.perf_counter() - t0, a.out), flush=True)
    if rank == 0:
        dt = time.perf_counter() - t0
        print("%s: %dx%d, %d spp%s on %d GPU(s) in %.2f s (%.1f Msamples/s) -> %s%s" % (
            caster.INTEGRATOR_NAMES[scene.integrator], scene.width, scene.height, a.spp,
            " on top of %d resumed" % base if base else "", world, dt,
            scene.width * scene.height * a.spp / dt / 1e6, a.out, " (tonemapped)" if tonemapped else ""))
    r.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
