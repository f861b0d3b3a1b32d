"""How one image is split over the GPUs of a node, and how the films meet again.

The reference is single-GPU (no NCCL/MPI anywhere in CUDA-PathTracer/tracer/);
what makes sharding possible is that a sample (pixel, iteration) is a pure
function of its seed, ``WangHash(pixel) + WangHash(iter)``
(CUDA-PathTracer/tracer/pathtracer.cu:881), and of the read-only scene.  One
process drives one GPU; the scene is replicated, and there is no data-path
collective until the single film exchange at the end:

``spp``   every rank renders the whole frame for its own, disjoint iterations
          (blocks of ``block`` iterations dealt round-robin); the films are
          summed onto rank 0 (``torch.distributed.reduce``, RCCL over xGMI on
          GPUs).  The sum differs from a one-GPU render only by fp32
          re-association across ranks.
``tile``  every rank renders all iterations for its own scanlines -- slabs of
          8 dealt round-robin over the ranks (``row_stripes``), so that every
          rank sees a sample of the whole image, or one contiguous band
          (``row_band``); the other rows of its film stay zero, so the same
          sum is an exact gather: bit-identical to a one-GPU render.

Nothing here touches a device: the functions take and return plain numbers and
``torch`` tensors, so the same code runs over ``gloo`` on CPUs in the tests.
"""


def spp_blocks(first_iter, n_iter, block, rank, world):
    """Contiguous iteration ranges (first, count) of [first_iter, first_iter + n_iter) that `rank` renders.
    Iteration i (counted from 1, as the reference counts them) lies in block (i - 1) // block, and block b goes
    to rank b % world: who renders an iteration depends on its number alone, not on how the caller cuts its
    calls, so successive short calls (one iteration per Render, as the reference's frame loop makes them) rotate
    over the ranks."""
    if block <= 0 or world <= 0 or not (0 <= rank < world):
        raise ValueError("bad sharding arguments")
    out = []
    it = first_iter
    end = first_iter + n_iter
    while it < end:
        b = max(it - 1, 0) // block
        block_end = min(end, (b + 1) * block + 1)
        if b % world == rank:
            out.append((it, block_end - it))
        it = block_end
    return out


def row_band(height, rank, world, align=8):
    """Scanlines [y0, y1) of `rank`: contiguous bands whose edges are multiples of
    `align` (the renderer generates camera rays in 8x8 pixel tiles)."""
    if world <= 0 or not (0 <= rank < world) or height < 0:
        raise ValueError("bad sharding arguments")
    units = (height + align - 1) // align
    lo = (units * rank) // world
    hi = (units * (rank + 1)) // world
    return min(lo * align, height), min(hi * align, height)


def row_stripes(height, rank, world, slab=8):
    """Interleaved scanline slabs of `rank`: (y0, y1, rows, stride) for caster_set_row_stripes, i.e. the
    scanlines y0 <= y < y1 with (y - y0) % stride < rows.  Slab j (scanlines [j * slab, (j + 1) * slab)) goes
    to rank j % world.  `slab` is a multiple of 8: the renderer generates camera rays in 8x8 pixel tiles, and
    whole tile rows are what lets it give path slots to its own scanlines only."""
    if world <= 0 or not (0 <= rank < world) or height < 0 or slab <= 0 or slab % 8:
        raise ValueError("bad sharding arguments")
    y0 = min(rank * slab, height)
    return y0, height, slab, world * slab


def stripe_rows(height, y0, y1, rows, stride):
    """The scanlines of a row set, as a list (what the renderer traces for those four numbers)."""
    return [y for y in range(max(y0, 0), min(y1, height)) if (y - y0) % stride < rows]


def reduce_film(film, dist=None, dst=0):
    """Sums the per-rank accumulated films onto `dst` (in place on that rank).
    `film` is a flat float32 tensor of W*H*3 on the device the process group's
    backend works with.  With no process group the film is returned unchanged."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return film
    dist.reduce(film, dst)
    return film


def total_iterations(n_iter_per_rank, dist=None):
    """Sum over ranks of the iterations each rendered (needed to divide the summed film)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return n_iter_per_rank
    import torch
    t = torch.tensor([float(n_iter_per_rank)], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t)
    return int(round(float(t.item())))
