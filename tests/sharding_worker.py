"""One rank of tests/test_sharding_gloo.py (not collected by pytest)."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402

sh = importlib.import_module("cuda-pathtracer_amd.sharding")


def main():
    scene, mode, spp, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    o = ol.Oracle(scene)
    acc = np.zeros((o.height, o.width, 3), np.float32)
    n_mine = 0
    if mode == "spp":
        for first, n in sh.spp_blocks(1, spp, 2, rank, world):
            o.render(first, n, 1, acc=acc)
            n_mine += n
    elif mode == "stripes":
        rows = sh.stripe_rows(o.height, *sh.row_stripes(o.height, rank, world))
        for y in rows[::8]:   # slabs of 8 scanlines (the last may be shorter)
            o.render(1, spp, 1, acc=acc, rows=(y, min(y + 8, o.height)))
        n_mine = spp if rank == 0 else 0
    else:
        y0, y1 = sh.row_band(o.height, rank, world)
        o.render(1, spp, 1, acc=acc, rows=(y0, y1))
        n_mine = spp if rank == 0 else 0
    film = torch.from_numpy(acc.reshape(-1))
    sh.reduce_film(film, dist, 0)
    total = sh.total_iterations(n_mine, dist)
    if rank == 0:
        assert total == spp
        np.save(out, film.numpy().reshape(o.height, o.width, 3))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
