"""The per-GPU share of bench.py's N-GPU line on ONE GPU: shard 0 and shard N - 1 of the Cornell box at 1024 x 1024 x 256 N spp (N = 1, 2, 4, 8) -- every shard must render the 268 M
samples of the N = 1 line in the N = 1 line's time (weak scaling: what is left to a node is the film reduce).   python tools/weak_shards.py"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mitsuba_amd import _ffi, _abi as A, scene as S
from mitsuba_amd.integrator import Scene, PathHIP
ft = _ffi.gaussian_filter()
sc = Scene(S.cornell_box(1024, 1024, ft).desc())
integ = PathHIP(maxDepth=-1)
film = torch.zeros((1024, 1024, 5), dtype=torch.float32, device="cuda")
for n in (1, 2, 4, 8):
    for shard in sorted({0, n - 1}):
        integ.render_device(sc, film.data_ptr(), 256 * n, seed=0, shard_index=shard, shard_count=n)
        torch.cuda.synchronize(); t = time.time()
        ok = integ.render_device(sc, film.data_ptr(), 256 * n, seed=0, shard_index=shard, shard_count=n, flags=A.PHIP_FLAG_KERNEL_TIMING)
        torch.cuda.synchronize(); dt = time.time() - t
        st = integ.stats.as_dict()
        w = film[..., 4].sum().item()
        print(json.dumps({"N": n, "shard": shard, "spp": 256 * n, "ok": bool(ok), "samples": st["samples"], "expected": 1024 * 1024 * 256, "ms": round(dt * 1e3, 2), "fused_ms": round(st["fused_kernel_ms"], 2),
                          "iterations": st["iterations"], "film_weight_sum": w, "finite": bool(torch.isfinite(film).all().item())}))
