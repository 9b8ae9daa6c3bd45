"""Block sharding across the GPUs of one node + reduction of the (R,G,B,alpha,weight) film.

The reference ships finished ImageBlocks to the master, which merges them under a mutex
(src/librender/renderproc.cpp:142-149; across machines StreamBackend::sendWorkResult,
src/libcore/sched_remote.cpp:519-532).  Here every rank (one process per GPU) renders the blocks
whose index in the reference's spiral order (src/librender/imageproc.cpp:43-78) is congruent to
its rank, into a private full-frame buffer; one `reduce(SUM)` over RCCL (backend "nccl" on
ROCm) -- or gloo in the CPU tests -- merges them on rank 0.  Filter footprints that cross block
borders are handled by the sum, exactly like the reference's block borders.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialises torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # PHIP_DIST_BACKEND=gloo: the per-process protocol on device tensors without RCCL (gloo stages them through the host) -- how the two-rank test
            # runs on a box with ONE GPU, where a RCCL clique of two ranks on the same device is refused (tests/test_gpu_round2.py)
            backend = os.environ.get("PHIP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def reduce_film(film, dst=0):
    """In-place SUM-reduce of a (H,W,5) float32 tensor onto rank `dst` (ncclReduce on GPUs)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
        if film.is_cuda:
            # A "synchronous" collective of the nccl backend only makes torch's CURRENT STREAM wait for it; the host returns at once.  What follows here
            # runs on the library's own streams (phip_film_to_host: the merged frame's D2H; the next phip_render_device: it overwrites `film`), which
            # know nothing of torch's -- so the host waits for the reduce before it hands the buffer on.
            torch.cuda.current_stream(film.device).synchronize()
    return film


def max_over_ranks(value, device="cpu"):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([float(value)], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return float(value)


def sum_over_ranks(value, device="cpu"):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([float(value)], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())
    return float(value)


class Solo:
    """The same interface without any collective: what ONE rank uses when it renders a job on its own while the others wait
    (bench.py: the single-GPU rate of the multi-GPU job, measured by rank 0 before the timed region)."""
    @staticmethod
    def barrier():
        pass

    @staticmethod
    def reduce_film(film, dst=0):
        return film

    @staticmethod
    def max_over_ranks(value, device="cpu"):
        return float(value)

    @staticmethod
    def sum_over_ranks(value, device="cpu"):
        return float(value)
