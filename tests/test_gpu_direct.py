"""GPU parity tests for the `direct` integrator (MIDirectIntegrator, direct.cpp:149-312): the HIP path
through the C ABI against the CPU oracle on the same seeded inputs.  Same bar as the path tracer: per-sample radiance
bit-identical, developed image within 1e-3 relative L2."""
import ctypes as C

import numpy as np
import pytest

from conftest import rel_l2
from mitsuba_amd import _abi as A, scene as S
from test_gpu_parity import compare_render, gpu  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def direct(**kw):
    from mitsuba_amd.integrator import DirectHIP
    return dict(integrator=DirectHIP, **kw)


@pytest.mark.parametrize("e,b", [(1, 1), (4, 0), (0, 3), (3, 2), (2, 5), (1, 4)])
def test_direct_cornell_matches_oracle(gpu, oracle, gauss, e, b):
    """every split of emitter / BSDF samples: rounds with and without a BSDF ray, the shared round, E = 0 and B = 0"""
    desc = S.cornell_box(96, 96, gauss).desc()
    same, r = compare_render(gpu, oracle, desc, 4, min_identical=1.0, **direct(emitterSamples=e, bsdfSamples=b))
    print("direct %d/%d: identical %.6f rel L2 %.3e" % (e, b, same, r))


def test_direct_equals_path_cut_at_depth_two_on_the_gpu(gpu, gauss):
    """shadingSamples = 1 is MIPathTracer with maxDepth = 2, sample for sample (tests/test_oracle_direct.py explains why)"""
    from mitsuba_amd.integrator import Scene, PathHIP, DirectHIP, HDRFilm
    gs = Scene(S.cornell_box(128, 128, gauss).desc())
    film = HDRFilm(128, 128)
    d = DirectHIP(); assert d.render(gs, film, 8, flags=A.PHIP_FLAG_SAMPLE_BUFFER); sd = d.samples(gs, 8)
    p = PathHIP(maxDepth=2); assert p.render(gs, film, 8, flags=A.PHIP_FLAG_SAMPLE_BUFFER); sp = p.samples(gs, 8)
    assert np.array_equal(sd.view(np.uint32), sp.view(np.uint32))
    assert d.stats.closest_rays == p.stats.closest_rays and d.stats.samples == p.stats.samples


def test_direct_hide_emitters_strict_normals_and_big_scenes(gpu, oracle, gauss):
    desc = S.cornell_box(64, 64, gauss).desc()
    compare_render(gpu, oracle, desc, 4, min_identical=1.0, **direct(shadingSamples=2, hideEmitters=True))
    # smooth-shaded, rough conductors, dielectrics (delta lobes: no MIS against the emitter density)
    # (the atrium has exact-t ties between coplanar triangles that the BVH and the kd-tree resolve in different orders:
    #  a handful of samples differ, as for the path tracer -- same film size and sample count as that test)
    desc = S.atrium(240, 136, gauss).desc()
    same, r = compare_render(gpu, oracle, desc, 4, min_identical=0.999, **direct(emitterSamples=2, bsdfSamples=2))
    print("direct atrium 2/2: identical %.6f rel L2 %.3e" % (same, r))
    same, r = compare_render(gpu, oracle, desc, 4, min_identical=0.999, **direct(strictNormals=True))
    print("direct atrium strict: identical %.6f rel L2 %.3e" % (same, r))
    desc = S.glass_room(240, 136, gauss).desc()
    same, r = compare_render(gpu, oracle, desc, 4, min_identical=0.999, **direct(emitterSamples=1, bsdfSamples=3))
    print("direct glass 1/3: identical %.6f rel L2 %.3e" % (same, r))


def env_scene(gauss, env, res=(128, 80)):
    sb = S.SceneBuilder()
    floor = sb.diffuse((0.4, 0.45, 0.5))
    mats = [sb.diffuse((0.7, 0.3, 0.2)), sb.twosided(sb.diffuse((0.2, 0.6, 0.3))),
            sb.roughconductor(alpha=0.2, eta=S.CU_ETA, k=S.CU_K), sb.dielectric(1.5, 1.0)]
    sb.quad((-6, 0, -6), (6, 0, -6), (6, 0, 6), (-6, 0, 6), floor, facing=(0, 1, 0))
    for i, m in enumerate(mats):
        P, T, N = S.sphere_mesh((-3 + 2 * i, 0.8, 0.5 * (i % 2)), 0.8, 24, 12)
        sb.mesh(P, T, m, normals=N)
    sb.quad((-1, 4, -1), (1, 4, -1), (1, 4, 1), (-1, 4, 1), sb.diffuse((0, 0, 0)), facing=(0, -1, 0), radiance=(8, 8, 6))
    env(sb)
    sb.perspective((0, 3, -9), (0, 0.5, 0), (0, 1, 0), 40.0)
    sb.hdrfilm(res[0], res[1], gauss)
    return sb


def test_direct_environment_emitters_match_oracle(gpu, oracle, gauss):
    """camera rays and BSDF-sampled rays that leave the scene (direct.cpp:160-165, 286-294), environment NEE"""
    from test_oracle_path import _sky, _rot
    desc = env_scene(gauss, lambda sb: sb.constant((0.9, 1.0, 1.2), sampling_weight=0.7)).desc()
    compare_render(gpu, oracle, desc, 4, min_identical=0.999, **direct(emitterSamples=2, bsdfSamples=2))
    compare_render(gpu, oracle, desc, 4, min_identical=0.999, **direct(hideEmitters=True, strictNormals=True))
    rng = np.random.default_rng(5)
    tex = (_sky(256, 128) * rng.uniform(0.5, 1.5, (128, 256, 1))).astype(np.float32)
    desc = env_scene(gauss, lambda sb: sb.envmap(tex, scale=0.8, to_world=_rot((1, 0.3, 0.2), 70.0), pyramid=True)).desc()
    same, r = compare_render(gpu, oracle, desc, 4, min_identical=0.999, **direct(emitterSamples=3, bsdfSamples=2))
    print("direct envmap: identical %.6f rel L2 %.3e" % (same, r))


def test_direct_bitmap_textures_match_oracle(gpu, oracle, gauss):
    """every BSDF query at the camera vertex sees the filtered texture value (its.getBSDF(ray), direct.cpp:175)"""
    from test_oracle_path import sphere_uvs, checker
    rng = np.random.default_rng(11)
    noise = rng.uniform(0.05, 0.95, (96, 160, 3)).astype(np.float32)
    sb = S.SceneBuilder()
    t_floor = sb.bitmap(checker(256, 32), filter_type="ewa", uscale=6.0, vscale=6.0)
    t_tri = sb.bitmap(noise, filter_type="trilinear", wrap="mirror", wrap_v="clamp", uscale=2.0, uoffset=0.3)
    sb.quad((-8, 0, -8), (8, 0, -8), (8, 0, 8), (-8, 0, 8), sb.diffuse(texture=t_floor), facing=(0, 1, 0), uvs=True)
    for i, m in enumerate([sb.diffuse(texture=t_tri), sb.twosided(sb.diffuse(texture=t_tri)), sb.roughconductor(alpha=0.05, alpha_v=0.3, eta=S.CU_ETA, k=S.CU_K)]):
        P, T, N = S.sphere_mesh((-2 + 2 * i, 0.8, 0.4 * (i % 2)), 0.8, 24, 12)
        sb.mesh(P, T, m, normals=N, uvs=sphere_uvs(N))
    sb.quad((-2, 5, -2), (2, 5, -2), (2, 5, 2), (-2, 5, 2), sb.diffuse((0, 0, 0)), facing=(0, -1, 0), radiance=(10, 10, 9))
    sb.constant((0.4, 0.5, 0.7))
    sb.perspective((0, 3.5, -11), (0, 0.4, 0), (0, 1, 0), 42.0)
    sb.hdrfilm(128, 96, gauss)
    compare_render(gpu, oracle, sb.desc(), 4, min_identical=0.999, **direct(emitterSamples=2, bsdfSamples=2))


def test_direct_shards_passes_and_errors(gpu, oracle, phip, gauss, monkeypatch):
    from mitsuba_amd.integrator import Scene, DirectHIP, HDRFilm
    desc = S.cornell_box(96, 64, gauss).desc()
    gs = Scene(desc)
    integ = DirectHIP(emitterSamples=2, bsdfSamples=2)
    whole = HDRFilm(96, 64); assert integ.render(gs, whole, 4)
    parts = HDRFilm(96, 64)
    for i in range(3):
        assert integ.render(gs, parts, 4, shard_index=i, shard_count=3)
    assert rel_l2(parts.storage, whole.storage) < 1e-6
    # several passes (bounded sample buffer) give the same film
    monkeypatch.setenv("PHIP_MAX_PASS_SAMPLES", str(96 * 64))
    passes = HDRFilm(96, 64); assert integ.render(gs, passes, 4)
    monkeypatch.delenv("PHIP_MAX_PASS_SAMPLES")
    assert np.allclose(whole.storage, passes.storage, rtol=1e-5, atol=1e-6)
    with pytest.raises(RuntimeError):
        DirectHIP(emitterSamples=0, bsdfSamples=0)                    # Assert, direct.cpp:107
    p = A.default_render_params(spp=1, integrator=A.PHIP_INTEGRATOR_DIRECT, emitter_samples=0, bsdf_samples=0)
    block = np.zeros((64, 96, 5), np.float32)
    st = A.phip_stats()
    from mitsuba_amd import _ffi
    assert phip.phip_render(gs._h, C.byref(p), _ffi.fptr(block), C.byref(st)) != 0
    assert b"emitterSamples" in phip.phip_last_error()
    p = A.default_render_params(spp=1, integrator=7)
    assert phip.phip_render(gs._h, C.byref(p), _ffi.fptr(block), C.byref(st)) != 0
