"""White-box validation of the oracle's MIDirectIntegrator restatement (src/integrators/direct/direct.cpp:149-312;
SURVEY 8f row 4).  No reference vector pins `direct` either, so it is cross-checked against the path tracer
restatement (`direct` with one emitter and one BSDF sample is the path tracer cut at maxDepth = 2), against the
closed-form irradiance of a rectangle light, and through the unbiasedness of every emitter/BSDF sample split."""
import numpy as np
import pytest

from conftest import rel_l2
from mitsuba_amd import _abi as A, scene as S
from test_oracle_path import point_to_rect_form_factor


def direct_params(**kw):
    kw.setdefault("emitter_samples", 1)
    kw.setdefault("bsdf_samples", 1)
    return A.default_render_params(integrator=A.PHIP_INTEGRATOR_DIRECT, **kw)


def test_direct_equals_path_cut_at_depth_two(oracle, gauss):
    """shadingSamples = 1: the same two MIS-weighted estimators as MIPathTracer's first vertex, the same sample
    block of the ctr stream -> per-sample radiance agrees to the last bit (the .5/.5 sample fractions only scale
    both arguments of the power heuristic by a power of two)"""
    for hide in (0, 1):
        desc = S.cornell_box(48, 48, gauss).desc()
        sc = oracle.OracleScene(desc)
        _, sd, std = sc.render(direct_params(spp=8, hide_emitters=hide), want_samples=True)
        _, sp, stp = sc.render(A.default_render_params(spp=8, max_depth=2, hide_emitters=hide), want_samples=True)
        assert np.array_equal(sd, sp)
        assert std.samples == stp.samples and std.closest_rays == stp.closest_rays


def test_direct_with_glossy_and_specular_materials_equals_path(oracle, gauss):
    """samples whose first vertex has a smooth BSDF still agree to the last bit; at a Dirac first vertex the two integrators read
    different numbers (`path` takes the next pair in call order, `direct` its pre-generated BSDF array), so only the expectation agrees"""
    desc = S.glass_room(40, 40, gauss, detail=0.2).desc()
    sc = oracle.OracleScene(desc)
    fd, sd, _ = sc.render(direct_params(spp=64), want_samples=True)
    fp, sp, _ = sc.render(A.default_render_params(spp=64, max_depth=2), want_samples=True)
    same = np.all(sd == sp, axis=-1).mean()
    assert same > 0.8, same
    a, b = oracle.develop(fd), oracle.develop(fp)
    assert abs(a.mean() - b.mean()) / b.mean() < 0.02, (a.mean(), b.mean())


def rect_light_scene(gauss, rho=0.6, le=5.0, h=2.0, a=1.5):
    sb = S.SceneBuilder()
    floor = sb.diffuse((rho, rho, rho)); black = sb.diffuse((0, 0, 0))
    sb.quad((-50, 0, -50), (50, 0, -50), (50, 0, 50), (-50, 0, 50), floor, facing=(0, 1, 0))
    sb.quad((-a / 2, h, -a / 2), (a / 2, h, -a / 2), (a / 2, h, a / 2), (-a / 2, h, a / 2), black, facing=(0, -1, 0), radiance=(le, le, le))
    sb.perspective((3.0, 1.0, 0.0), (0.0, 0.0, 0.0), (0, 1, 0), 0.5, near=1e-2, far=100.0)
    sb.hdrfilm(8, 8, gauss)
    return sb, rho * le * 4 * point_to_rect_form_factor(a / 2, a / 2, h)


@pytest.mark.parametrize("e,b,spp,tol", [(1, 1, 2048, 0.01), (4, 0, 512, 0.01), (0, 4, 2048, 0.03), (3, 2, 512, 0.01), (1, 5, 512, 0.015)])
def test_direct_closed_form_for_every_sample_split(oracle, gauss, e, b, spp, tol):
    """L_o = rho * Le * F(dA -> light): emitter sampling alone, BSDF sampling alone and the MIS combinations
    (direct.cpp:130-138 weights) all converge to the same closed-form value"""
    sb, expect = rect_light_scene(gauss)
    sc = oracle.OracleScene(sb.desc())
    film, _, st = sc.render(direct_params(spp=spp, emitter_samples=e, bsdf_samples=b))
    got = oracle.develop(film).mean()
    assert abs(got - expect) / expect < tol, (got, expect)
    assert st.samples == 64 * spp
    assert st.closest_rays <= st.samples * (1 + b)


def test_direct_sfmt_stream_agrees_with_ctr_stream(oracle, gauss):
    """`independent` semantics (sample arrays drawn per pixel by Sampler::generate) against the parity stream"""
    desc = S.cornell_box(32, 32, gauss).desc()
    sc = oracle.OracleScene(desc)
    for e, b in ((1, 1), (2, 3)):
        a = oracle.develop(sc.render(direct_params(spp=128, emitter_samples=e, bsdf_samples=b))[0])
        s = oracle.develop(sc.render(direct_params(spp=128, emitter_samples=e, bsdf_samples=b), sampler="sfmt", threads=2)[0])
        assert rel_l2(a, s) < 0.08
        assert abs(a.mean() - s.mean()) / a.mean() < 0.02


def test_direct_environment_and_strict_normals(oracle, gauss):
    """camera rays that miss see the environment unless hideEmitters (direct.cpp:160-165); BSDF-sampled rays that
    miss are weighted against the environment's direct-sampling density (direct.cpp:286-294)"""
    sb = S.SceneBuilder()
    m = sb.diffuse((0.5, 0.5, 0.5))
    sb.quad((-1, 0, -1), (1, 0, -1), (1, 0, 1), (-1, 0, 1), m, facing=(0, 1, 0))
    sb.constant((1.0, 1.0, 1.0))
    sb.perspective((0, 1.0, 3.0), (0, 0.2, 0), (0, 1, 0), 45.0, near=1e-2, far=100.0)
    sb.hdrfilm(32, 32, gauss)
    sc = oracle.OracleScene(sb.desc())
    img = oracle.develop(sc.render(direct_params(spp=256, emitter_samples=2, bsdf_samples=2))[0])
    film = sc.render(direct_params(spp=64, hide_emitters=1))[0]
    hid = oracle.develop(film)
    alpha = film[..., 3] / film[..., 4]
    floor = alpha > 0.999
    sky = alpha == 0
    assert floor.sum() > 50 and sky.sum() > 50
    assert np.allclose(img[sky], 1.0, atol=1e-5) and np.allclose(hid[sky], 0.0, atol=2e-3)
    # an unoccluded diffuse plane under a uniform white sky: L = rho
    assert abs(img[floor].mean() - 0.5) < 0.01
    assert abs(hid[floor].mean() - 0.5) < 0.02
    # strictNormals changes nothing on flat-shaded geometry
    s1 = sc.render(direct_params(spp=4, strict_normals=1), want_samples=True)[1]
    s0 = sc.render(direct_params(spp=4), want_samples=True)[1]
    assert np.array_equal(s0, s1)


def test_direct_validation(oracle, gauss):
    sc = oracle.OracleScene(S.cornell_box(8, 8, gauss).desc())
    with pytest.raises(RuntimeError):
        sc.render(direct_params(spp=1, emitter_samples=0, bsdf_samples=0))      # Assert, direct.cpp:107
    with pytest.raises(RuntimeError):
        sc.render(direct_params(spp=1, emitter_samples=-1))
