"""White-box validation of the oracle's Li / emitter sampling / film (no reference test
pins these): furnace test, closed-form direct lighting of a rectangle light, depth semantics,
the reference's SFMT streams against the counter-based parity stream, libm against phip_fmath."""
import numpy as np
import pytest

from conftest import rel_l2
from mitsuba_amd import _abi as A, scene as S


def furnace_scene(gauss, rho=0.5, le=1.0, n=24):
    """closed unit cube whose six faces are area emitters with a diffuse BRDF: L = Le / (1 - rho) everywhere"""
    sb = S.SceneBuilder()
    m = sb.diffuse((rho, rho, rho))
    c = [0, 1]
    faces = [((0, 0, 0), (1, 0, 0), (1, 0, 1), (0, 0, 1), (0, 1, 0)), ((0, 1, 0), (1, 1, 0), (1, 1, 1), (0, 1, 1), (0, -1, 0)),
             ((0, 0, 0), (0, 1, 0), (0, 1, 1), (0, 0, 1), (1, 0, 0)), ((1, 0, 0), (1, 1, 0), (1, 1, 1), (1, 0, 1), (-1, 0, 0)),
             ((0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1)), ((0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1), (0, 0, -1))]
    for p0, p1, p2, p3, f in faces:
        sb.quad(p0, p1, p2, p3, m, facing=f, radiance=(le, le, le))
    sb.perspective((0.5, 0.5, 0.2), (0.45, 0.55, 0.9), (0, 1, 0), 70.0, near=1e-3, far=10.0)
    sb.hdrfilm(n, n, gauss)
    return sb


def test_furnace_radiance_is_le_over_one_minus_rho(oracle, gauss):
    """NEE + hit-emitter MIS + Russian roulette are jointly unbiased: every pixel converges to Le/(1-rho)"""
    sb = furnace_scene(gauss, rho=0.5, le=1.0)
    sc = oracle.OracleScene(sb.desc())
    film, _, st = sc.render(A.default_render_params(spp=256, max_depth=-1, rr_depth=5))
    rgb = oracle.develop(film)
    assert abs(rgb.mean() - 2.0) < 0.02, rgb.mean()
    assert np.allclose(film[..., 3] / film[..., 4], 1.0, atol=1e-5)          # alpha = 1 inside a closed box
    # truncated series: maxDepth = D keeps emission + D-1 scattering orders: Le * sum_{i<D} rho^i
    # (maxDepth=1: emitters only, 2: direct illumination only -- integrator.cpp:199-201)
    film4, _, _ = sc.render(A.default_render_params(spp=128, max_depth=4))
    assert abs(oracle.develop(film4).mean() - (1 + 0.5 + 0.25 + 0.125)) < 0.02
    film2, _, _ = sc.render(A.default_render_params(spp=128, max_depth=2))
    assert abs(oracle.develop(film2).mean() - 1.5) < 0.02
    assert st.path_vertices / st.samples > 3


def point_to_rect_form_factor(a, b, h):
    """differential area under the CORNER of an a x b rectangle at height h (parallel planes)"""
    A_, B_ = np.sqrt(a * a + h * h), np.sqrt(b * b + h * h)
    return (a / A_ * np.arctan(b / A_) + b / B_ * np.arctan(a / B_)) / (2 * np.pi)


def test_direct_lighting_closed_form(oracle, gauss):
    """maxDepth=2 (direct only): L_o = rho * Le * F(dA -> light) under the centre of a square light"""
    rho, le, h, a = 0.6, 5.0, 2.0, 1.5
    sb = S.SceneBuilder()
    floor = sb.diffuse((rho, rho, rho)); black = sb.diffuse((0, 0, 0))
    sb.quad((-50, 0, -50), (50, 0, -50), (50, 0, 50), (-50, 0, 50), floor, facing=(0, 1, 0))
    sb.quad((-a / 2, h, -a / 2), (a / 2, h, -a / 2), (a / 2, h, a / 2), (-a / 2, h, a / 2), black, facing=(0, -1, 0), radiance=(le, le, le))
    sb.perspective((3.0, 1.0, 0.0), (0.0, 0.0, 0.0), (0, 1, 0), 0.5, near=1e-2, far=100.0)    # narrow view of the point below the light
    sb.hdrfilm(8, 8, gauss)
    sc = oracle.OracleScene(sb.desc())
    film, _, _ = sc.render(A.default_render_params(spp=2048, max_depth=2))
    got = oracle.develop(film).mean()
    expect = rho * le * 4 * point_to_rect_form_factor(a / 2, a / 2, h)
    assert abs(got - expect) / expect < 0.01, (got, expect)
    # full MIS estimator (maxDepth=3 adds one bounce off a floor that sees only the black light back side: same value)
    film3, _, _ = sc.render(A.default_render_params(spp=1024, max_depth=3))
    assert abs(oracle.develop(film3).mean() - expect) / expect < 0.02


def test_depth_and_emitter_visibility_semantics(oracle, gauss):
    """path.cpp:135,148-165: maxDepth=1 -> directly visible emitters only; hideEmitters removes exactly those"""
    desc = S.cornell_box(64, 64, gauss).desc()
    sc = oracle.OracleScene(desc)
    d1 = oracle.develop(sc.render(A.default_render_params(spp=8, max_depth=1))[0])
    assert d1.max() > 3.9 and (d1 > 0).mean() < 0.05            # only the light's pixels
    lit = d1[..., 0] > 1.0
    dfull = oracle.develop(sc.render(A.default_render_params(spp=32, max_depth=4))[0])
    dhide = oracle.develop(sc.render(A.default_render_params(spp=32, max_depth=4, hide_emitters=1))[0])
    inner = np.zeros_like(lit); inner[2:-2, 2:-2] = True
    core = lit & inner
    assert dhide[core].mean() < 0.2 * dfull[core].mean()
    near = d1[..., 0] > 0                                          # anything the light's samples touch through the filter
    for _ in range(2):
        near = near | np.roll(near, 1, 0) | np.roll(near, -1, 0) | np.roll(near, 1, 1) | np.roll(near, -1, 1)
    assert rel_l2(dhide[~near & inner], dfull[~near & inner]) < 1e-6   # same sample stream elsewhere
    d2 = oracle.develop(sc.render(A.default_render_params(spp=32, max_depth=2))[0])
    assert d2[~lit].mean() < dfull[~lit].mean()                # direct-only is darker than 3 bounces
    with pytest.raises(RuntimeError, match="maxDepth"):
        sc.render(A.default_render_params(spp=1, max_depth=0))
    with pytest.raises(RuntimeError, match="rrDepth"):
        sc.render(A.default_render_params(spp=1, rr_depth=0))


def test_sfmt_streams_agree_statistically_with_ctr_stream(oracle, gauss):
    """`independent` semantics (one SFMT19937 clone per worker, sequential consumption) and the
    counter-based parity stream estimate the same image"""
    desc = S.cornell_box(48, 48, gauss).desc()
    sc = oracle.OracleScene(desc)
    p = A.default_render_params(spp=192, max_depth=5)
    a = oracle.develop(sc.render(p, threads=4, sampler="sfmt")[0])
    b = oracle.develop(sc.render(p, threads=4, sampler="ctr")[0])
    c = oracle.develop(sc.render(A.default_render_params(spp=192, max_depth=5, seed=7), sampler="ctr")[0])
    pool = lambda x: x.reshape(6, 8, 6, 8, 3).mean(axis=(1, 3))
    assert rel_l2(pool(a), pool(b)) < 0.03
    assert rel_l2(pool(c), pool(b)) < 0.03
    assert abs(a.mean() / b.mean() - 1) < 0.01
    # one worker: the SFMT render is reproducible
    a1 = sc.render(p, threads=1, sampler="sfmt")[0]; a2 = sc.render(p, threads=1, sampler="sfmt")[0]
    assert (a1 == a2).all()


def test_libm_and_phip_fmath_modes_agree(oracle, gauss):
    """substituting include/phip_fmath.h for libm changes a render by far less than the parity tolerance"""
    oracle.build(libm=True)
    sb = S.cornell_box(64, 64, gauss)
    cu = sb.twosided(sb.roughconductor(S.CU_ETA, S.CU_K, alpha=0.2))
    P, T, N = S.sphere_mesh((185, 240, 170), 70.0, 24, 12); sb.mesh(P, T, cu, normals=N)
    desc = sb.desc()
    p = A.default_render_params(spp=16, max_depth=6)
    pm, sm, _ = oracle.OracleScene(desc).render(p, want_samples=True)
    lm, sl, _ = oracle.OracleScene(desc, libm=True).render(p, want_samples=True)
    assert oracle.lib(True).oracle_uses_libm() == 1 and oracle.lib(False).oracle_uses_libm() == 0
    r = rel_l2(oracle.develop(lm), oracle.develop(pm))
    assert r < 1e-3, r
    close = np.isclose(sm, sl, rtol=1e-3, atol=1e-5).all(axis=-1)
    assert close.mean() > 0.995                                  # a few paths flip a branch on the last ulp


def test_oracle_shards_sum_to_whole_and_threads_do_not_matter(oracle, gauss):
    desc = S.cornell_box(100, 70, gauss).desc()
    sc = oracle.OracleScene(desc)
    whole = sc.render(A.default_render_params(spp=4, max_depth=4), threads=3)[0]
    again = sc.render(A.default_render_params(spp=4, max_depth=4), threads=1)[0]
    assert (whole == again).all()                                # ctr stream + fixed merge order: deterministic
    acc = np.zeros_like(whole)
    for r in range(4):
        acc += sc.render(A.default_render_params(spp=4, max_depth=4, shard_index=r, shard_count=4))[0]
    assert rel_l2(acc, whole) < 1e-6


def test_film_weights_and_filter_table(oracle, phip, gauss):
    """rfilter.cpp:38-57: 31 normalised samples of the Gaussian + a trailing zero; the product computes the same bits"""
    from mitsuba_amd import _ffi
    r, t = gauss
    assert r == 2.0 and len(t) == 32 and t[31] == 0.0
    assert all(t[i] >= t[i + 1] for i in range(31))
    assert abs(sum(t) * 2 * r / 31 - 1.0) < 1e-5
    assert _ffi.gaussian_filter(0.5) == oracle.gaussian_filter(0.5)
    assert _ffi.gaussian_filter(0.7) == oracle.gaussian_filter(0.7)
    # interior pixels receive total weight ~ spp (the discretised filter integrates to ~1 over the pixel lattice)
    sc = oracle.OracleScene(S.cornell_box(40, 40, gauss).desc())
    film = sc.render(A.default_render_params(spp=64, max_depth=2))[0]
    w = film[4:-4, 4:-4, 4]
    assert abs(w.mean() / 64 - 1.0) < 0.05
