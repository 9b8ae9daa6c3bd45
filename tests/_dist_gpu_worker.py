"""Worker for tests/test_gpu_round2.py::test_two_ranks_on_one_gpu_run_the_per_process_protocol: bench.py's N > 1 step -- phip_render_device of the rank's
shard into a device film, mitsuba_amd.distributed.reduce_film onto rank 0, phip_film_to_host of the merged frame -- by two (or more) processes that share
GPU 0, with gloo carrying the device tensors through the host (PHIP_DIST_BACKEND=gloo).  What is under test is the ORDER of the three actors on one buffer:
the library's stream that renders into `film`, torch / gloo that reduce it, the library's stream that copies it out and overwrites it in the next step
(the bug class of 5f696ea: the host did not wait for the reduce before the library read the film)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from mitsuba_amd import _abi as A, _ffi, scene as S, distributed as D  # noqa: E402
from mitsuba_amd.integrator import Scene, PathHIP, PinnedFilm         # noqa: E402


def main(out_path, steps):
    rank, world, local = D.init_from_env()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    W, H, spp = 640, 360, 8
    desc = S.cornell_box(W, H, _ffi.gaussian_filter(0.5)).desc()
    scene = Scene(desc, device=local)
    integ = PathHIP(maxDepth=5)
    film = torch.zeros((H, W, 5), dtype=torch.float32, device=dev)
    host = PinnedFilm(W, H) if rank == 0 else None
    frames = []
    for step in range(steps):
        assert integ.render_device(scene, film.data_ptr(), spp, seed=step, shard_index=rank, shard_count=world)
        D.reduce_film(film, dst=0)
        if rank == 0:
            scene.film_to_host(film.data_ptr(), host.ptr)
            frames.append(host.storage.copy())
    total = D.sum_over_ranks(integ.stats.samples, dev)
    if rank == 0:
        errs = []
        whole = PinnedFilm(W, H)
        for step in range(steps):                                   # the unsharded frame of the same seed
            assert integ.render_into(scene, whole.ptr, spp, seed=step)
            w = whole.storage
            errs.append(float(np.linalg.norm(frames[step].astype(np.float64) - w) / np.linalg.norm(w)))
        with open(out_path, "w") as f:
            f.write("%d %d %.9g %d\n" % (world, steps, max(errs), int(total)))
    D.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
