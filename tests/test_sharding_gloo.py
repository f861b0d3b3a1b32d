"""The N > 1 path on CPUs: two processes over gloo split one image by iterations
(spp), by scanline bands (tile) and by interleaved slabs of scanlines (stripes) with the product's sharding plans and film
reduce; the CPU oracle stands in for the GPU renderer (no GPU here).  Tile
sharding must be bit-identical to one process, spp sharding within fp32
re-association (relRMSE <= 1e-5)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
WORKER = os.path.join(ROOT, "tests", "sharding_worker.py")


@pytest.mark.parametrize("mode", ["spp", "tile", "stripes"])
def test_two_ranks_match_one(tmp_path, mode):
    scene = os.path.join(GOLD, "scenes", "cornell_pt", "scene.json")
    spp = 6
    out = str(tmp_path / "film.npy")
    port = 29500 + (os.getpid() % 2000) + ["spp", "tile", "stripes"].index(mode)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, WORKER, scene, mode, str(spp), out], env=env))
    for p in procs:
        assert p.wait(timeout=240) == 0
    got = np.load(out)
    want = ol.Oracle(scene).render(1, spp, 2)
    if mode != "spp":
        np.testing.assert_array_equal(got, want)
    else:
        d = (got - want) / (np.abs(want) + 1e-2)
        assert float(np.sqrt(np.mean(d * d))) <= 1e-5
