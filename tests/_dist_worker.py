"""Worker for tests/test_distributed.py: one process per rank (gloo on CPU).  The renderer behind
the shard is the CPU oracle here (there is no GPU in this container); what is under test is the
N>1 protocol of bench.py: spiral-order block sharding + mitsuba_amd.distributed.reduce_film."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from mitsuba_amd import _abi as A, scene as S, distributed as D  # noqa: E402
from oracle import oracle_ffi as O  # noqa: E402


def main(out_path):
    rank, world, local = D.init_from_env(backend="gloo")
    gauss = O.gaussian_filter(0.5)
    desc = S.cornell_box(96, 64, gauss).desc()
    sc = O.OracleScene(desc)
    p = A.default_render_params(spp=4, max_depth=4, shard_index=rank, shard_count=world)
    film, _, st = sc.render(p, threads=2)
    t = torch.from_numpy(film.copy())
    D.barrier()
    D.reduce_film(t, dst=0)
    total = D.sum_over_ranks(st.samples)
    slowest = D.max_over_ranks(float(rank + 1))
    if rank == 0:
        whole, _, st_all = sc.render(A.default_render_params(spp=4, max_depth=4), threads=2)
        err = float(np.linalg.norm(t.numpy() - whole) / np.linalg.norm(whole))
        with open(out_path, "w") as f:
            f.write("%d %d %.9g %d %d %g %g\n" % (world, rank, err, int(total), int(st_all.samples), slowest, float(film[..., 4].sum())))
    D.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
