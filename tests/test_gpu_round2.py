"""GPU tests of the round-2 additions, through the C ABI: the fused kernel's dispatch, axis-aligned rays (zero direction
components in the slab test), multi-device rendering inside phip_render (one host thread per device + film merge),
progressive passes (sample_offset / PHIP_FLAG_ACCUMULATE), the sticky cancellation flag and the progress callback."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import rel_l2
from mitsuba_amd import _abi as A, scene as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(phip):
    if phip.phip_device_count() <= 0:
        pytest.fail("no HIP device visible: " + phip.phip_last_error().decode())
    from mitsuba_amd import integrator
    return integrator


def test_fused_kernel_is_chosen_for_lds_resident_scenes_and_trees_that_live_in_l2(gpu, gauss):
    cb = gpu.Scene(S.cornell_box(64, 64, gauss).desc())
    assert cb.accel_info().fits_lds == 1
    integ = gpu.PathHIP(maxDepth=5)
    film = gpu.HDRFilm(64, 64)
    assert integ.render(cb, film, 4)
    assert integ.stats.fused == 1 and integ.stats.iterations == 1 and integ.stats.n_devices == 1
    assert integ.render(cb, gpu.HDRFilm(64, 64), 4, flags=A.PHIP_FLAG_NO_FUSED)
    assert integ.stats.fused == 0 and integ.stats.iterations > 1
    d = gpu.DirectHIP()
    assert d.render(cb, gpu.HDRFilm(64, 64), 4) and d.stats.fused == 1      # round 6: `direct` rides the fused kernel too (k_mega<.., DIRECT>)
    assert d.render(cb, gpu.HDRFilm(64, 64), 4, flags=A.PHIP_FLAG_NO_FUSED) and d.stats.fused == 0
    for nu, nv, fits in ((4, 3, 1), (12, 8, 0)):
        sb = S.cornell_box(64, 64, gauss)
        P, T, N = S.sphere_mesh((200, 300, 200), 45.0, nu, nv)
        sb.mesh(P, T, sb.dielectric(1.33, 1.0), normals=N)
        glass = gpu.Scene(sb.desc())
        # round 5: dielectric / microfacet models ride the fused kernel on the packed leaf table (<= 64 Wald records); round 6: beyond it the scene gets the compressed
        # 8-wide tree and the fused kernel walks that from memory (fused_traversal 4) -- the wavefront kernels only with PHIP_FLAG_NO_FUSED, where they are k_shade + k_rays_w
        ai = glass.accel_info()
        assert ai.fits_lds == fits and (fits or (ai.fused_traversal == 4 and ai.node_bytes == 80)), (len(T), ai.as_dict())
        assert integ.render(glass, gpu.HDRFilm(64, 64), 4) and integ.stats.fused == 1
        assert integ.render(glass, gpu.HDRFilm(64, 64), 4, flags=A.PHIP_FLAG_NO_FUSED) and integ.stats.fused == 0 and integ.stats.iterations > 1
        glass.close()
    big = gpu.Scene(S.atrium(64, 36, gauss).desc())
    assert big.accel_info().fits_lds == 0


def axis_rays(rng, n, lo, hi):
    """rays along +-X / +-Y / +-Z (incl. -0.0 components) and rays with one or two zero components"""
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = np.zeros((n, 3), np.float32)
    kind = rng.integers(0, 3, n)
    for i in range(n):
        if kind[i] == 0:                         # one axis
            d[i, rng.integers(0, 3)] = rng.choice([-1.0, 1.0])
        elif kind[i] == 1:                       # one zero component
            v = rng.normal(size=3); v[rng.integers(0, 3)] = 0.0
            d[i] = v / np.linalg.norm(v)
        else:                                    # axis direction with negative zeros elsewhere
            a = rng.integers(0, 3); d[i] = -0.0; d[i, a] = rng.choice([-1.0, 1.0])
    rays = np.zeros((n, 8), np.float32)
    rays[:, :3] = o; rays[:, 3] = 1e-4; rays[:, 4:7] = d; rays[:, 7] = np.inf
    return rays


def test_axis_aligned_rays_on_an_origin_centred_scene(gpu, oracle, gauss):
    """a zero direction component must not turn the slab test into NaNs: boxes straddling 0, origin != 0"""
    rng = np.random.default_rng(11)
    n = 4000
    c = rng.uniform(-4, 4, (n, 1, 3))
    p = (c + rng.normal(scale=0.5, size=(n, 3, 3))).astype(np.float32).reshape(-1, 3)
    idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    sb = S.SceneBuilder()
    sb.mesh(p, idx, sb.diffuse((0.5, 0.5, 0.5)))
    sb.perspective((0, 0, -20), (0, 0, 0), (0, 1, 0), 45.0)
    sb.hdrfilm(32, 32, gauss)
    desc = sb.desc()
    gs = gpu.Scene(desc); osc = oracle.OracleScene(desc)
    rays = axis_rays(rng, 30000, -5, 5)
    gh, go, _ = gs.rayIntersect(rays, True, True)
    oh, oo, _ = osc.trace(rays, True, True)
    assert (oh[:, 3].view(np.uint32) != A.PHIP_NO_HIT).mean() > 0.3          # the workload does hit things
    same = (gh.view(np.uint32) == oh.view(np.uint32)).all(axis=1)
    assert same.mean() > 0.9995, same.mean()                                 # (exact-t ties on shared edges aside)
    assert (go == oo).mean() > 0.9995
    # the Cornell box as well (walls are axis-aligned: rays run inside wall planes' boxes)
    desc = S.cornell_box(32, 32, gauss).desc()
    gs = gpu.Scene(desc); osc = oracle.OracleScene(desc)
    rays = axis_rays(rng, 30000, 0, 550)
    gh, go, _ = gs.rayIntersect(rays, True, True)
    oh, oo, _ = osc.trace(rays, True, True)
    assert ((gh.view(np.uint32) == oh.view(np.uint32)).all(axis=1)).mean() > 0.9995
    assert (go == oo).mean() > 0.9995


@pytest.mark.parametrize("scene,md", [("cornell_box", 5), ("atrium", 4)])
def test_multi_device_render_equals_the_single_device_frame(gpu, phip, gauss, scene, md):
    """n_devices > 1 through the C ABI: one host thread + stream per device, blocks dealt in spiral order, films merged on
    devices[0].  On a box with >= 2 GPUs the merge is ncclReduce; a 1-GPU box lists the GPU twice (alias test hook)."""
    w, h, spp = (160, 96, 4)
    desc = getattr(S, scene)(w, h, gauss).desc()
    gs = gpu.Scene(desc)
    integ = gpu.PathHIP(maxDepth=md)
    one = gpu.HDRFilm(w, h)
    assert integ.render(gs, one, spp)
    samples = integ.stats.samples
    ngpu = phip.phip_device_count()
    cases = [([0, 0], A.PHIP_FLAG_ALIAS_DEVICES), ([0, 0, 0], A.PHIP_FLAG_ALIAS_DEVICES)]
    if ngpu >= 2:
        cases.append((list(range(min(ngpu, 2))), 0))
    for devs, fl in cases:
        multi = gpu.HDRFilm(w, h)
        assert integ.render(gs, multi, spp, flags=fl, devices=devs)
        assert integ.stats.n_devices == len(devs) and integ.stats.samples == samples
        # same samples, same per-pixel weights; only the order of the float additions across devices differs
        assert rel_l2(multi.storage, one.storage) < 2e-6
        assert np.abs(multi.storage[..., 4] - one.storage[..., 4]).max() < 1e-4
    # a second call on the same replicas, sharded from outside as well (2 ranks x 2 devices)
    parts = gpu.HDRFilm(w, h)
    for r in range(2):
        assert integ.render(gs, parts, spp, shard_index=r, shard_count=2, flags=A.PHIP_FLAG_ALIAS_DEVICES, devices=[0, 0])
    assert rel_l2(parts.storage, one.storage) < 2e-6
    # error behaviour
    with pytest.raises(Exception):
        integ.render(gs, gpu.HDRFilm(w, h), spp, devices=[0, 0])            # listed twice without the alias flag
    with pytest.raises(Exception):
        integ.render(gs, gpu.HDRFilm(w, h), spp, devices=[0, 99], flags=A.PHIP_FLAG_ALIAS_DEVICES)


def test_eight_device_aliases_progress_reduce_and_cancel(gpu, phip, gauss):
    """the in-library multi-device path as an eight-GPU node drives it -- eight host threads, eight streams, eight private films merged on
    devices[0] -- on the one GPU of this box listed eight times (PHIP_FLAG_ALIAS_DEVICES): the merged frame is the single-device frame, reduce_ms is reported, the
    progress callback is entered by all eight device threads (one at a time) and never reports more than the job holds, and a cancel request in the middle of the
    frame ends all eight renders (renderproc.cpp:142-154: the process returns, the frame is not delivered) and leaves the scene usable."""
    import threading
    w, h, spp = 256, 160, 8
    gs = gpu.Scene(S.cornell_spheres(w, h, gauss).desc())              # the fused kernel on the wide tree, eight ways
    integ = gpu.PathHIP(maxDepth=6)
    one = gpu.HDRFilm(w, h)
    assert integ.render(gs, one, spp)
    samples = integ.stats.samples
    devs = [0] * 8
    seen = {}
    lock = threading.Lock(); inside = [0]; overlap = [0]

    def progress(user, device, done, total):
        with lock:
            inside[0] += 1
            overlap[0] = max(overlap[0], inside[0])
        seen.setdefault(threading.get_ident(), []).append((device, done, total))
        with lock:
            inside[0] -= 1

    multi = gpu.HDRFilm(w, h)
    assert integ.render(gs, multi, spp, flags=A.PHIP_FLAG_ALIAS_DEVICES, devices=devs, progress=progress)
    st = integ.stats
    assert st.n_devices == 8 and st.samples == samples and st.reduce_ms > 0
    assert rel_l2(multi.storage, one.storage) < 2e-6
    assert np.abs(multi.storage[..., 4] - one.storage[..., 4]).max() < 1e-4
    assert len(seen) == 8, len(seen)                                   # every device thread reported
    assert overlap[0] == 1                                             # ... one at a time (phip_scene::progressLock)
    for calls in seen.values():
        assert all(0 <= d <= t for _, d, t in calls) and calls[-1][1] == calls[-1][2]      # a device's last report is its whole share
    assert sum(calls[-1][2] for calls in seen.values()) == samples
    # cancel in the middle of a long frame: all eight threads stop, the call reports the cancellation, the next render is whole again
    big = gpu.Scene(S.cornell_spheres(1024, 1024, gauss).desc())
    integ2 = gpu.PathHIP(maxDepth=-1)
    film = gpu.HDRFilm(1024, 1024)
    assert integ2.render(big, film, 1, flags=A.PHIP_FLAG_ALIAS_DEVICES, devices=devs)
    integ2._scene = big
    t = threading.Timer(0.1, integ2.cancel); t.start()
    ok = integ2.render(big, film, 1024, flags=A.PHIP_FLAG_ALIAS_DEVICES, devices=devs)     # ~1 G samples, eight shards one after another on one GPU: ~0.5 s
    t.join()
    assert ok is False
    again = gpu.HDRFilm(1024, 1024); ref = gpu.HDRFilm(1024, 1024)
    assert integ2.render(big, again, 2, flags=A.PHIP_FLAG_ALIAS_DEVICES, devices=devs) is True
    assert integ2.render(big, ref, 2) is True
    assert rel_l2(again.storage, ref.storage) < 2e-6
    big.close(); gs.close()


def test_scene_replicate(gpu, phip, gauss):
    gs = gpu.Scene(S.cornell_box(32, 32, gauss).desc())
    gs.replicate([0])
    if phip.phip_device_count() >= 2:
        gs.replicate([0, 1])
    with pytest.raises(Exception):
        gs.replicate([5, 0])                                                # devices[0] must be the scene's device


@pytest.mark.parametrize("fused", [True, False])
def test_progressive_passes_add_up_to_the_single_render(gpu, phip, gauss, fused):
    """sample_offset / sample_total / PHIP_FLAG_ACCUMULATE: 3 + 5 samples in two calls = one 8-sample render (same sample
    indices, hence the same radiance values; the film differs by the order of the float additions only)"""
    w, h = 96, 64
    gs = gpu.Scene(S.cornell_box(w, h, gauss).desc())
    integ = gpu.PathHIP(maxDepth=6)
    base = 0 if fused else A.PHIP_FLAG_NO_FUSED
    whole = gpu.HDRFilm(w, h)
    assert integ.render(gs, whole, 8, flags=base)
    block = np.zeros((h, w, 5), np.float32)
    st = A.phip_stats()
    p = integ.params(gs, 3, flags=base, sample_offset=0, sample_total=8)
    assert phip.phip_render(gs._h, C.byref(p), block.ctypes.data_as(C.POINTER(C.c_float)), C.byref(st)) == 0
    first = block.copy()
    p = integ.params(gs, 5, flags=base | A.PHIP_FLAG_ACCUMULATE, sample_offset=3, sample_total=8)
    assert phip.phip_render(gs._h, C.byref(p), block.ctypes.data_as(C.POINTER(C.c_float)), C.byref(st)) == 0
    assert st.samples == w * h * 5
    assert rel_l2(block, whole.storage) < 2e-6
    assert np.abs(block[..., 4] - whole.storage[..., 4]).max() < 1e-4
    assert (block[..., 4] > first[..., 4]).all()
    # sample_offset + spp beyond sample_total is an error
    p = integ.params(gs, 5, flags=base, sample_offset=4, sample_total=8)
    assert phip.phip_render(gs._h, C.byref(p), block.ctypes.data_as(C.POINTER(C.c_float)), C.byref(st)) == A.PHIP_ERR_INVALID


@pytest.mark.parametrize("fused", [True, False])
def test_cancel_is_sticky_and_consumed(gpu, gauss, fused):
    """a phip_cancel that arrives before the render starts cancels that render (Scheduler::cancel semantics); the render
    that observed it consumes the flag"""
    gs = gpu.Scene(S.cornell_box(64, 64, gauss).desc())
    integ = gpu.PathHIP(maxDepth=5)
    fl = 0 if fused else A.PHIP_FLAG_NO_FUSED
    film = gpu.HDRFilm(64, 64)
    assert integ.render(gs, film, 2, flags=fl)
    integ.cancel()
    film2 = gpu.HDRFilm(64, 64)
    assert integ.render(gs, film2, 64, flags=fl) is False
    assert (film2.storage == 0).all()
    film3 = gpu.HDRFilm(64, 64)
    assert integ.render(gs, film3, 2, flags=fl)
    assert (film3.storage.view(np.uint32) == film.storage.view(np.uint32)).all()


def test_progress_callback_and_cancel_from_it(gpu, gauss):
    w, h, spp = 128, 128, 32
    gs = gpu.Scene(S.atrium(w, h, gauss).desc())
    integ = gpu.PathHIP(maxDepth=6)
    seen = []
    assert integ.render(gs, gpu.HDRFilm(w, h), spp, progress=lambda user, dev, done, total: seen.append((dev, done, total)))
    assert seen and seen[-1][1] == seen[-1][2] == w * h * spp
    assert all(a[1] <= b[1] for a, b in zip(seen, seen[1:]))
    # cancelling from inside the callback stops the render
    calls = []

    def stop(user, dev, done, total):
        calls.append(done)
        integ.cancel()
    assert integ.render(gs, gpu.HDRFilm(w, h), 256, progress=stop) is False
    assert len(calls) >= 1 and calls[0] < w * h * 256


@pytest.mark.timeout(900)
@pytest.mark.parametrize("launcher", ["torchrun", "in_library"])
def test_bench_two_gpus(gpu, phip, launcher):
    """bench.py --gpus 2 on a box with at least two GPUs (skipped otherwise): one process per GPU over RCCL (the driver's launch line), and the
    library's own multi-device path (host threads + ncclReduce from C++).  One JSON line, n_gpus 2, every sample of the job rendered."""
    import json, socket, subprocess, sys
    if phip.phip_device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = ["bench.py", "--gpus", "2", "--workload", "cornell_256x256_16spp_md4", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"]
    if launcher == "torchrun":
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port)] + args
    else:
        cmd = [sys.executable] + args
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=840, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["workload"] == "cornell_256x256_16spp_md4"


@pytest.mark.timeout(900)
def test_bench_default_line_of_two_ranks_sharing_one_gpu(gpu, phip):
    """The line the driver's SCALE run asks for -- `bench.py --gpus 2` with no --workload under two ranks -- executed end to end on a box with ONE GPU: both ranks on
    GPU 0, the film reduced by gloo through the host (PHIP_DIST_BACKEND=gloo; a RCCL clique of two ranks on one device is refused).  `value` is the metric's config at
    2 x its samples per pixel (weak scaling: every rank renders the 268 M samples of the N = 1 line), and the fixed 4K job of BASELINE.json configs[4] rides along
    under "workloads" with the one-GPU rate of the same job.  (Rates of two ranks that share a GPU mean nothing; the counts and the keys are what is checked.)"""
    import json, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PHIP_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=root, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=840) for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[0][-1500:] + o[1][-3000:] for o in outs)
    d = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["workload"] == "cornell_1024x1024_512spp" and d["config"]["spp"] == 512
    assert "scaling_efficiency" not in d and d["cpu_baseline"] is None
    assert d["roofline"]["traffic_source"] and "cornell_1024x1024_256spp" in d["roofline"]["traffic_source"]     # a rank's launch is the N = 1 config's launch
    job = d["workloads"]["atrium_3840x2160_1024spp_md8"]
    assert job["scaling"] == "strong" and job["value"] > 0 and job["single_gpu_same_job"]["value"] > 0
    assert job["spp"] == 1024 and job["width"] == 3840


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 3])
def test_two_ranks_on_one_gpu_run_the_per_process_protocol(gpu, phip, tmp_path, world):
    """bench.py's N > 1 step (phip_render_device of the rank's shard -> reduce_film -> phip_film_to_host on rank 0) executed by `world`
    processes that SHARE GPU 0, the device tensors reduced by gloo through the host (a RCCL clique of two ranks on one device is refused): the order of the
    library's streams and the collective on one film buffer -- the bug class of 5f696ea, found by reading in round 4 -- is run by a machine.  Three steps back
    to back (the next render overwrites the buffer the previous step reduced and copied out); every merged frame equals the unsharded render of its seed."""
    import socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "result.txt"
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PHIP_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "tests", "_dist_gpu_worker.py"), str(out), "3"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=540)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    w, steps, err, total = out.read_text().split()
    assert int(w) == world and int(steps) == 3
    assert float(err) < 1e-6, err                               # shards merged in another order of additions than the unsharded film: last bits only
    assert int(total) == 640 * 360 * 8                          # every sample of the (last) frame rendered by exactly one rank


def test_rccl_calls_of_the_merge_on_the_devices_that_are_there(gpu, phip):
    """ncclCommInitAll / ncclGroupStart / ncclReduce / ncclGroupEnd as renderMultiDevice issues them, bound through the same dlopen,
    on a clique of the visible devices (one device is a valid clique: the only way these calls run on a single-GPU box)"""
    phip.phip_debug_rccl_selftest.argtypes = [C.c_int, C.c_size_t]
    n = phip.phip_debug_rccl_selftest(8, 1 << 20)
    assert n >= 1, phip.phip_last_error()
    assert n == min(8, phip.phip_device_count())


def test_frame_reaches_pinned_and_pageable_host_memory_alike(gpu, phip, gauss):
    """ABI 7: phip_render delivers the frame into pinned memory (phip_host_alloc: one asynchronous copy) and into pageable memory (staged chunks) -- the same
    bits, a film larger than the 4 MB staging chunk incl. a ragged last chunk; phip_film_to_host does the same for a device-resident frame (what the
    per-process multi-GPU path copies after its reduce); stats.d2h_ms is reported inside render_ms"""
    import torch
    from mitsuba_amd.integrator import PinnedFilm
    w, h = 701, 333                                               # 701 x 333 x 20 B = 4.67 MB: two chunks, the second ragged
    sc = gpu.Scene(S.cornell_box(w, h, gauss).desc()); integ = gpu.PathHIP(maxDepth=4)
    ref = gpu.HDRFilm(w, h); assert integ.render(sc, ref, 2)
    pf = PinnedFilm(w, h); assert integ.render_into(sc, pf.ptr, 2)
    assert integ.stats.d2h_ms > 0 and integ.stats.render_ms >= integ.stats.d2h_ms
    pageable = np.zeros((h, w, 5), np.float32); assert integ.render_into(sc, pageable.ctypes.data, 2)
    assert (pf.storage.view(np.uint32) == ref.storage.view(np.uint32)).all() and (pageable.view(np.uint32) == ref.storage.view(np.uint32)).all()
    dev = torch.zeros((h, w, 5), dtype=torch.float32, device="cuda:0")
    assert integ.render_device(sc, dev.data_ptr(), 2)
    out = np.zeros((h, w, 5), np.float32); sc.film_to_host(dev.data_ptr(), out.ctypes.data)
    pf.storage[...] = 0; sc.film_to_host(dev.data_ptr(), pf.ptr)
    assert (out.view(np.uint32) == ref.storage.view(np.uint32)).all() and (pf.storage.view(np.uint32) == ref.storage.view(np.uint32)).all()
    pf.close(); sc.close()
