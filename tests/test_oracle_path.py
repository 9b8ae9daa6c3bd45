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


def test_constant_environment_closed_form(oracle, gauss):
    """constant.cpp + path.cpp:136-143,233-265: a convex diffuse surface under a uniform environment L radiates
    exactly rho * L (emitter sampling and BSDF sampling both draw cosine-distributed directions, so every
    sample carries 0.5 rho L + 0.5 rho L); the background is L; hideEmitters removes only the background"""
    rho, L = np.array([0.5, 0.25, 0.75], np.float32), np.array([1.0, 2.0, 3.0], np.float32)
    sb = S.SceneBuilder()
    m = sb.diffuse(tuple(rho))
    sb.quad((-1, -1, 0), (1, -1, 0), (1, 1, 0), (-1, 1, 0), m, facing=(0, 0, -1))
    sb.constant(tuple(L))
    sb.perspective((0, 0, -6), (0, 0, 0), (0, 1, 0), 45.0)
    sb.hdrfilm(48, 48, gauss)
    sc = oracle.OracleScene(sb.desc())
    film, smp, st = sc.render(A.default_render_params(spp=8, max_depth=-1), want_samples=True)
    img = oracle.develop(film)
    on_quad = film[..., 3] / film[..., 4] > 0.999          # alpha == 1: every sample of the pixel hit the quad
    off_quad = film[..., 3] == 0
    assert on_quad.sum() > 100 and off_quad.sum() > 500
    assert np.abs(img[on_quad] / (rho * L) - 1).max() < 1e-5
    assert np.abs(img[off_quad] / L - 1).max() < 1e-5
    # per sample: alpha 0 -> L, alpha 1 -> rho L (zero variance)
    hit = smp[..., 3] == 1
    assert np.abs(smp[..., :3][hit] / (rho * L) - 1).max() < 1e-5 and (smp[..., :3][~hit] == L).all()
    assert st.path_vertices == st.samples                   # one surface vertex at most (Li's depth counter starts at 1)
    # hideEmitters: path.cpp:139-141 drops the directly visible environment only
    fh = sc.render(A.default_render_params(spp=8, max_depth=-1, hide_emitters=1))[0]
    ih = oracle.develop(fh)
    assert (ih[off_quad] == 0).all() and np.abs(ih[on_quad] / (rho * L) - 1).max() < 1e-5
    # maxDepth = 1: emitted radiance only; maxDepth = 2 already holds the whole (single-bounce) transport
    i1 = oracle.develop(sc.render(A.default_render_params(spp=8, max_depth=1))[0])
    assert (i1[on_quad] == 0).all() and np.abs(i1[off_quad] / L - 1).max() < 1e-5
    i2 = oracle.develop(sc.render(A.default_render_params(spp=8, max_depth=2))[0])
    assert np.abs(i2[on_quad] / (rho * L) - 1).max() < 1e-5
    # the back of a one-sided diffuse surface is black; a twosided one is lit from both sides
    # (refN = 0 for BSDFs with a back side: the emitter is then sampled over the whole sphere, constant.cpp:193-196)
    sb2 = S.SceneBuilder()
    one = sb2.diffuse(tuple(rho)); two = sb2.twosided(sb2.diffuse(tuple(rho)))
    sb2.quad((-2.2, -1, 0), (-0.2, -1, 0), (-0.2, 1, 0), (-2.2, 1, 0), one, facing=(0, 0, 1))
    sb2.quad((0.2, -1, 0), (2.2, -1, 0), (2.2, 1, 0), (0.2, 1, 0), two, facing=(0, 0, 1))
    sb2.constant(tuple(L))
    sb2.perspective((0, 0, -8), (0, 0, 0), (0, 1, 0), 45.0)
    sb2.hdrfilm(64, 32, gauss)
    sc2 = oracle.OracleScene(sb2.desc())
    f2 = sc2.render(A.default_render_params(spp=256, max_depth=-1))[0]
    i2 = oracle.develop(f2)
    solid = f2[..., 3] / f2[..., 4] > 0.999
    left = solid & (np.arange(64)[None, :] < 32); right = solid & (np.arange(64)[None, :] >= 32)
    assert left.sum() > 30 and right.sum() > 30
    if i2[left].max() > 0: left, right = right, left        # (the image x axis runs against world x for this camera)
    assert i2[left].max() < 0.01 and np.median(i2[left]) == 0      # (border pixels see a trace of the background through the filter)
    assert np.abs(i2[right].mean(axis=0) / (rho * L) - 1).max() < 0.02      # uniform-sphere NEE: no longer zero variance


def test_environment_validation(oracle, gauss):
    sb = S.SceneBuilder(); sb.diffuse((0.5, 0.5, 0.5))
    sb.constant((1, 1, 1)); sb.constant((2, 2, 2))
    sb.perspective((0, 0, -5), (0, 0, 0), (0, 1, 0), 45.0); sb.hdrfilm(8, 8, gauss)
    with pytest.raises(RuntimeError, match="one environment emitter"):      # scene.cpp:510-513
        oracle.OracleScene(sb.desc())
    # an empty scene under an environment: every pixel shows it
    sb = S.SceneBuilder(); sb.diffuse((0.5, 0.5, 0.5)); sb.constant((0.25, 0.5, 1.0))
    sb.perspective((0, 0, -5), (0, 0, 0), (0, 1, 0), 45.0); sb.hdrfilm(8, 8, gauss)
    img = oracle.develop(oracle.OracleScene(sb.desc()).render(A.default_render_params(spp=2))[0])
    assert np.abs(img / np.array([0.25, 0.5, 1.0]) - 1).max() < 1e-5


def _sky(w=64, h=32):
    """procedural HDR lat-long map: blue-ish gradient, warm ground, a bright sun patch"""
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    t = (y + 0.5) / h
    sky = np.stack([0.2 + 0.3 * t, 0.3 + 0.4 * t, 0.9 - 0.5 * t], -1)
    ground = np.stack([0.25 + 0 * t, 0.2 + 0 * t, 0.15 + 0 * t], -1)
    tex = np.where((t < 0.5)[..., None], sky, ground).astype(np.float32)
    sun = ((x - 0.7 * w) ** 2 + (y - 0.25 * h) ** 2) < (0.04 * w) ** 2
    tex[sun] = (60.0, 50.0, 35.0)
    return np.ascontiguousarray(tex)


def _rot(axis, deg):
    a = np.radians(deg); c, s_ = np.cos(a), np.sin(a)
    x, y, z = np.asarray(axis, np.float64) / np.linalg.norm(axis)
    R = np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * s_, x * z * (1 - c) + y * s_],
                  [y * x * (1 - c) + z * s_, c + y * y * (1 - c), y * z * (1 - c) - x * s_],
                  [z * x * (1 - c) - y * s_, z * y * (1 - c) + x * s_, c + z * z * (1 - c)]])
    m = np.eye(4); m[:3, :3] = R
    return m.astype(np.float32)


def test_envmap_illumination(oracle, gauss):
    """envmap.cpp: (1) a uniform map behaves like the constant emitter (rho * L on a convex diffuse surface);
    (2) importance sampling and BSDF sampling estimate the same irradiance (MIS consistency: maxDepth=2 image from
    emitter sampling alone == from both strategies, statistically); (3) toWorld rotates the illumination;
    (4) directly visible pixels need hideEmitters or the explicit bilinear-background flag"""
    rho, L = np.array([0.5, 0.25, 0.75], np.float32), np.array([1.0, 2.0, 3.0], np.float32)

    def quad_scene(env, res=32):
        sb = S.SceneBuilder(); m = sb.diffuse(tuple(rho))
        sb.quad((-1, -1, 0), (1, -1, 0), (1, 1, 0), (-1, 1, 0), m, facing=(0, 0, -1))
        env(sb)
        sb.perspective((0, 0, -6), (0, 0, 0), (0, 1, 0), 45.0); sb.hdrfilm(res, res, gauss)
        return sb
    # (1)
    sc = oracle.OracleScene(quad_scene(lambda sb: sb.envmap(np.tile(L, (16, 32, 1)).astype(np.float32))).desc())
    f = sc.render(A.default_render_params(spp=512, max_depth=-1, hide_emitters=1))[0]
    on = f[..., 3] / f[..., 4] > 0.999
    assert np.abs(oracle.develop(f)[on].mean(axis=0) / (rho * L) - 1).max() < 0.01
    assert (oracle.develop(f)[f[..., 3] == 0] == 0).all()                 # hidden background
    with pytest.raises(RuntimeError, match="hideEmitters"):
        sc.render(A.default_render_params(spp=1, max_depth=-1))
    fb = sc.render(A.default_render_params(spp=2, max_depth=-1, flags=A.PHIP_FLAG_ENVMAP_BILINEAR_BACKGROUND))[0]
    assert np.abs(oracle.develop(fb)[fb[..., 3] == 0] / L - 1).max() < 1e-5
    # (2) + (3): sun map; the quad faces -z.  Reference irradiance by brute-force quadrature over the map.
    tex = _sky()
    h, w, _ = tex.shape

    def irradiance(n, R):
        th = (np.arange(h) + 0.5) * np.pi / h; ph = (np.arange(w) + 0.5) * 2 * np.pi / w
        T, P = np.meshgrid(th, ph, indexing="ij")
        d = np.stack([np.sin(P) * np.sin(T), np.cos(T), -np.cos(P) * np.sin(T)], -1) @ R[:3, :3].T      # envmap.cpp:596-598, then toWorld
        cosn = np.clip(d @ np.asarray(n, np.float64), 0, None)
        dw = np.sin(T) * (np.pi / h) * (2 * np.pi / w)
        return (tex.astype(np.float64) * (cosn * dw)[..., None]).sum(axis=(0, 1))
    for R in (np.eye(4, dtype=np.float32), _rot((0, 1, 0), 140.0), _rot((1, 0.3, 0.2), 70.0)):
        sc = oracle.OracleScene(quad_scene(lambda sb: sb.envmap(tex, scale=0.5, to_world=R), res=16).desc())
        f = sc.render(A.default_render_params(spp=4096, max_depth=-1, hide_emitters=1))[0]
        on = f[..., 3] / f[..., 4] > 0.999
        got = oracle.develop(f)[on].mean(axis=0)
        expect = rho / np.pi * 0.5 * irradiance((0, 0, -1), R)
        assert np.abs(got / expect - 1).max() < 0.03, (got, expect)        # texel-centre quadrature vs bilinear map


def test_envmap_filtered_background(oracle, gauss):
    """camera rays carry differentials: envmap.cpp:395-407 -> MIPMap::eval (EWA).  (1) a constant map stays constant under
    any filter; (2) on a noisy high-resolution map seen through a coarse film the filtered background has the same mean
    as the unfiltered one and less pixel-to-pixel variance; (3) more samples per pixel shrink the footprint
    (scaleDifferential by 1/sqrt(spp), integrator.cpp:144-145); (4) level sizes follow mipmap.h:182-192"""
    assert [l.shape[:2] for l in S.mip_pyramid(np.zeros((5, 12, 3), np.float32))] == [(5, 12), (3, 6), (2, 3), (1, 2), (1, 1)]
    L = np.array([1.0, 2.0, 3.0], np.float32)

    def scene(tex, pyramid, res=32, fov=70.0):
        sb = S.SceneBuilder(); sb.diffuse((0.5, 0.5, 0.5))
        sb.envmap(tex, pyramid=pyramid)
        sb.perspective((0, 0, -6), (0, 0, 0), (0, 1, 0), fov); sb.hdrfilm(res, res, gauss)
        return oracle.OracleScene(sb.desc())
    img = oracle.develop(scene(np.tile(L, (256, 512, 1)).astype(np.float32), True).render(A.default_render_params(spp=1))[0])
    assert np.abs(img / L - 1).max() < 1e-5
    rng = np.random.default_rng(3)
    tex = (_sky(1024, 512) * rng.uniform(0.5, 1.5, (512, 1024, 1))).astype(np.float32)
    ewa = oracle.develop(scene(tex, True).render(A.default_render_params(spp=1))[0])
    bil = oracle.develop(scene(tex, False).render(A.default_render_params(spp=1, flags=A.PHIP_FLAG_ENVMAP_BILINEAR_BACKGROUND))[0])
    hi = lambda a: np.abs(np.diff(a, axis=1)).mean()                 # pixel-to-pixel roughness
    assert abs(ewa.mean() / bil.mean() - 1) < 0.01 and hi(ewa) < 0.75 * hi(bil)
    ewa64 = oracle.develop(scene(tex, True).render(A.default_render_params(spp=64))[0])
    bil64 = oracle.develop(scene(tex, False).render(A.default_render_params(spp=64, flags=A.PHIP_FLAG_ENVMAP_BILINEAR_BACKGROUND))[0])
    assert np.abs(ewa64 - bil64).mean() < 0.5 * np.abs(ewa - bil).mean()      # smaller footprint: closer to the unfiltered lookup
    with pytest.raises(RuntimeError, match="pyramid"):
        scene(tex, False).render(A.default_render_params(spp=1))


def sphere_uvs(N):
    """latitude-longitude texture coordinates of unit normals (u wraps at the seam: a few triangles span it -- fine here)"""
    N = np.asarray(N, np.float64)
    return np.stack([np.mod(np.arctan2(N[:, 2], N[:, 0]) / (2 * np.pi), 1.0), np.arccos(np.clip(N[:, 1], -1, 1)) / np.pi], -1).astype(np.float32)


def checker(n=64, cells=8, lo=0.1, hi=0.9):
    t = np.full((n, n, 3), lo, np.float32)
    t[(np.indices((n, n)).sum(0) // (n // cells)) % 2 == 0] = hi
    return t * np.array([1.0, 0.8, 0.6], np.float32)


def test_bitmap_texture(oracle, gauss):
    """`bitmap` reflectance on a diffuse BSDF (bitmap.cpp, texture.cpp:112-121, mipmap.h) with UV
    tangents (trimesh.cpp:683-735) and first-vertex UV partials (intersection.cpp:5-76).  A convex diffuse surface under
    a unit constant environment radiates its albedo, so the image shows the texture itself."""
    def scene(tex_kw, res=64, fov=40.0, uvs=True, spp_scene=None):
        sb = S.SceneBuilder()
        tid = None if tex_kw is None else sb.bitmap(**tex_kw)
        m = sb.diffuse((0.5, 0.25, 0.75), texture=tid)
        sb.quad((-1, -1, 0), (1, -1, 0), (1, 1, 0), (-1, 1, 0), m, facing=(0, 0, -1), uvs=uvs)
        sb.constant((1.0, 1.0, 1.0))
        sb.perspective((0, 0, -4), (0, 0, 0), (0, 1, 0), fov); sb.hdrfilm(res, res, gauss)
        return oracle.OracleScene(sb.desc())
    P = A.default_render_params
    # a constant texture is the constant reflectance
    c = np.tile(np.array([0.5, 0.25, 0.75], np.float32), (8, 8, 1))
    a = oracle.develop(scene(dict(image=c)).render(P(spp=4, max_depth=3))[0])
    b = oracle.develop(scene(None).render(P(spp=4, max_depth=3))[0])
    assert np.abs(a - b).max() < 1e-6
    # two-texel texture, nearest filter: left half / right half of the quad (u runs with the quad's first edge)
    two = np.array([[[0.2, 0.2, 0.2], [0.8, 0.8, 0.8]]], np.float32)
    sc = scene(dict(image=two, filter_type="nearest"), res=64, fov=25.0)       # the quad fills the frame
    f = sc.render(P(spp=4, max_depth=2))[0]
    img = oracle.develop(f)[..., 0]
    left, right = img[:, 4:28].mean(), img[:, 36:60].mean()
    assert {round(float(left), 3), round(float(right), 3)} == {0.2, 0.8}
    # bilinear: a linear ramp between the texel centres (u = 0.25 .. 0.75), wrap mode decides the outer quarters
    ramp = lambda wrap: oracle.develop(scene(dict(image=two, filter_type="bilinear", wrap=wrap), res=64, fov=25.0).render(P(spp=16, max_depth=2))[0])[32, :, 0]
    rc, rr = ramp("clamp"), ramp("repeat")
    if rc[10] > rc[50]: rc, rr = rc[::-1], rr[::-1]
    assert abs(rc[2] - 0.2) < 5e-3 and abs(rc[61] - 0.8) < 5e-3 and abs(rc[31] + rc[32] - 1.0) < 0.02      # clamped ends, symmetric ramp
    assert 0.3 < rr[1] < 0.5 < rr[62] < 0.7 and abs(rr[1] + rr[62] - 1.0) < 0.02                    # repeat blends 0.8 <-> 0.2 across the edge (the frame shows u = 0.06 .. 0.94)
    # uscale / uoffset (Texture2D): three repetitions
    rep = oracle.develop(scene(dict(image=two, filter_type="nearest", uscale=3.0), res=96, fov=25.0).render(P(spp=4, max_depth=2))[0])[48, :, 0]
    assert (np.abs(np.diff((rep > 0.5).astype(int))).sum()) == 5
    # filtered first vertex: a fine checkerboard seen through a coarse film aliases without the EWA lookup
    chk = checker(256, 64)
    noisy = oracle.develop(scene(dict(image=chk, filter_type="bilinear"), res=24, fov=25.0).render(P(spp=1, max_depth=2))[0])[4:20, 4:20, 0]
    ewa = oracle.develop(scene(dict(image=chk, filter_type="ewa"), res=24, fov=25.0).render(P(spp=1, max_depth=2))[0])[4:20, 4:20, 0]
    assert abs(ewa.mean() / 0.5 - 1) < 0.05 and ewa.std() < 0.5 * noisy.std()
    # texture coordinates change dpdu and with it the shading frame (skdtree.h:373-380) -- but not a diffuse image
    a = oracle.develop(scene(None, uvs=True).render(P(spp=64, max_depth=3))[0])
    b = oracle.develop(scene(None, uvs=None).render(P(spp=64, max_depth=3))[0])
    assert abs(a.mean() / b.mean() - 1) < 0.01
    # energy conservation is enforced like diffuse.cpp:95
    with pytest.raises(RuntimeError, match="ensureEnergyConservation"):
        scene(dict(image=two * 2))


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


def test_sweep_mode_equals_the_kd_tree_on_ordinary_rays(oracle, gauss):
    """the oracle's test hook that answers ray queries by testing every triangle (o_kdtree.h: bruteForce) renders the same bits as
    the kd-tree restatement wherever no ray grazes a split plane (tests/test_ref_pin.py pins the one place it does not)"""
    for desc, spp, md in ((S.cornell_box(64, 64, gauss).desc(), 16, -1), (S.glass_room(32, 18, gauss, detail=0.1).desc(), 4, 8)):
        sc = oracle.OracleScene(desc)
        p = A.default_render_params(spp=spp, max_depth=md)
        f0, s0, st0 = sc.render(p, want_samples=True)
        sc.set_bruteforce(True)
        f1, s1, st1 = sc.render(p, want_samples=True)
        same = (s0.view(np.uint32) == s1.view(np.uint32)).all(-1)
        assert same.mean() > 0.9999, same.mean()
        assert st0.samples == st1.samples
        sc.close()


def test_ld_sampler_reduces_the_error_of_smooth_integrands(oracle, gauss):
    """PHIP_SAMPLER_LD against the counter stream at equal sample counts: on a scene whose pixel integrals are smooth in the first
    dimensions (pixel filter, one bounce of an area light on diffuse walls) the stratified (0,2)-sequences give a clearly lower
    error against a converged image -- the reason ldsampler exists -- and the same expectation"""
    desc = S.cornell_box(24, 24, gauss).desc()
    sc = oracle.OracleScene(desc)
    ref_film, _, _ = sc.render(A.default_render_params(spp=4096, max_depth=2, seed=77))
    ref = oracle.develop(ref_film)
    err = {}
    for name, smp in (("ctr", A.PHIP_SAMPLER_CTR), ("ld", A.PHIP_SAMPLER_LD)):
        e = []
        for seed in range(4):
            f, _, _ = sc.render(A.default_render_params(spp=16, max_depth=2, seed=seed, sampler=smp))
            e.append(np.mean((oracle.develop(f) - ref) ** 2))
        err[name] = float(np.mean(e))
    assert err["ld"] < 0.6 * err["ctr"], err
    big, _, _ = sc.render(A.default_render_params(spp=1024, max_depth=2, seed=5, sampler=A.PHIP_SAMPLER_LD))
    assert abs(oracle.develop(big).mean() - ref.mean()) / ref.mean() < 0.01
    # error behaviour: a sample count that is not a power of two, the `direct` integrator
    with pytest.raises(RuntimeError):
        sc.render(A.default_render_params(spp=12, sampler=A.PHIP_SAMPLER_LD))
    # `direct` with sample arrays: same expectation, lower error than the counter stream
    dref = oracle.develop(sc.render(A.default_render_params(spp=2048, integrator=A.PHIP_INTEGRATOR_DIRECT, emitter_samples=1, bsdf_samples=1, seed=9))[0])
    derr = {}
    for name, smp in (("ctr", A.PHIP_SAMPLER_CTR), ("ld", A.PHIP_SAMPLER_LD)):
        e = [np.mean((oracle.develop(sc.render(A.default_render_params(spp=4, integrator=A.PHIP_INTEGRATOR_DIRECT, emitter_samples=4, bsdf_samples=2,
                                                                              seed=s, sampler=smp))[0]) - dref) ** 2) for s in range(4)]
        derr[name] = float(np.mean(e))
    assert derr["ld"] < 0.7 * derr["ctr"], derr
    sc.close()


def test_every_c2_sample_that_differs_from_the_kd_tree_is_a_kd_tree_artifact(oracle, gauss):
    """profiles/r02_c2_fullsize_sample_parity.json lists the 20 samples of BASELINE's C2 (268 M) on which the GPU and the oracle's kd-tree
    (= Mitsuba's, bit for bit) disagreed, with both radiance values as measured on the MI355X.  Re-evaluated here on the CPU, one path
    each, with the stream keys of the full frame: the oracle with its kd-tree reproduces the value recorded for the reference, the
    oracle answering ray queries by a sweep over every triangle reproduces the value the GPU computed -- every single difference is
    the kd-tree returning another closest hit (silhouette edge lost at a split plane, or an exact-distance tie decided by leaf order),
    none is an arithmetic difference."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r02_c2_fullsize_sample_parity.json")
    rec = json.load(open(path))["gpu_vs_oracle_kd_tree"]["samples"]
    assert len(rec) == 20
    sc = oracle.OracleScene(S.cornell_box(1024, 1024, gauss).desc())
    p = A.default_render_params(spp=16, sample_total=256)
    for r in rec:
        sc.set_bruteforce(False)
        kd = sc.path_sample(p, r["x"], r["y"], r["k"])
        sc.set_bruteforce(True)
        sw = sc.path_sample(p, r["x"], r["y"], r["k"])
        assert np.array_equal(kd, np.array(r["oracle"], np.float32)), r
        assert np.array_equal(sw, np.array(r["gpu"], np.float32)), r
        assert not np.array_equal(kd, sw)
    sc.close()


def test_environment_emitters_sample_direct_matches_pdf_chi_square(oracle, gauss):
    """The reference's test_chisquare::test03_EmitterDirect (test_chisquare.cpp:575-618; data/tests/test_emitter.xml holds environment emitters only):
    chi-square (10 x 20 bins in (theta, phi), significance 0.01 / number of tests) that Emitter::sampleDirect draws directions with the density
    pdfDirect reports, from a reference point at the origin -- for the `envmap` emitter (hierarchical row / column CDFs + tent within the texel,
    envmap.cpp:516-556, 573-644; a smooth map and a rotated one) and for `constant` (uniform sphere)."""
    import ctypes as C
    from test_oracle_bsdf import chi2_test, SIGNIFICANCE
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    L = oracle.lib()
    h, w = 32, 64
    T, P = np.meshgrid((np.arange(h) + 0.5) / h * np.pi, (np.arange(w) + 0.5) / w * 2 * np.pi, indexing="ij")
    lum = 1.0 + 0.7 * np.cos(2 * P) * np.sin(T) + 0.5 * np.cos(T) + 2.5 * np.exp(-((T - 0.9) ** 2 + (P - 2.0) ** 2) / 0.18)
    tex = np.stack([lum, 0.8 * lum, 0.6 * lum], -1).astype(np.float32)
    c, s = np.cos(0.7), np.sin(0.7)
    R = np.array([[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1]], np.float32) @ np.array([[1, 0, 0, 0], [0, 0.8, -0.6, 0], [0, 0.6, 0.8, 0], [0, 0, 0, 1]], np.float32)
    cases = [("envmap", lambda sb: sb.envmap(tex)), ("envmap rotated", lambda sb: sb.envmap(tex, scale=0.5, to_world=R)), ("constant", lambda sb: sb.constant((1.0, 2.0, 3.0)))]
    rng = np.random.default_rng(2024)
    n = 400000
    for name, env in cases:
        sb = S.SceneBuilder(); m = sb.diffuse((0.5, 0.5, 0.5))
        sb.quad((-1, -1, 0), (1, -1, 0), (1, 1, 0), (-1, 1, 0), m, facing=(0, 0, -1))
        env(sb)
        sb.perspective((0, 0, -6), (0, 0, 0), (0, 1, 0), 45.0); sb.hdrfilm(8, 8, gauss)
        sc = oracle.OracleScene(sb.desc())
        ref = np.zeros(3, np.float32)
        smp = np.minimum(rng.random((n, 2)).astype(np.float32), np.float32(1) - np.float32(2 ** -24))
        d = np.zeros((n, 3), np.float32); pdf = np.zeros(n, np.float32); val = np.zeros((n, 3), np.float32)
        assert L.oracle_env_sample_direct(sc.h, fp(ref), n, fp(smp), fp(d), fp(pdf), fp(val)) == 0

        def pdf_fn(dirs):
            out = np.zeros(len(dirs), np.float32)
            assert L.oracle_env_pdf_direct(sc.h, fp(ref), len(dirs), fp(np.ascontiguousarray(dirs, np.float32)), fp(out)) == 0
            return out
        pval, mass = chi2_test(d, pdf_fn, n)
        assert pval > SIGNIFICANCE / len(cases), (name, pval)
        assert abs(mass - (pdf > 0).mean()) < 1e-2, (name, mass)                     # the density integrates to the fraction of successful samples
        ok = pdf > 0
        # sampleDirect's pdf is pdfDirect of the direction it returns (up to the round trip direction -> (u, v) -> texel weights near texel borders and the poles)
        rel = np.abs(pdf_fn(d[ok]) - pdf[ok]) / pdf[ok]
        assert np.quantile(rel, 0.999) < 1e-3, (name, np.quantile(rel, 0.999), rel.max())    # (at the poles 1 / sin(theta) makes single samples differ by more)
        sc.close()
