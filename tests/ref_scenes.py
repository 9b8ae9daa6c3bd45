"""The scenes on which the oracle is pinned to the REFERENCE ITSELF (oracle/_ref = /root/reference compiled in place).

Used three ways:
  tests/test_ref_pin.py          live: reference vs oracle (libm build, the reference's SFMT sampler stream), where
                                 /root/reference or a prebuilt oracle/_ref exists
  tests/golden/make_golden_ref.py  writes the reference's per-sample radiance (and the MIP pyramids its code built)
                                 to tests/golden/ref_renders.npz
  tests/test_golden.py           checks the oracle against that fixture anywhere (no reference needed)
Every scene is small enough to render in well under a second on either side."""
import numpy as np

from mitsuba_amd import _abi as A, scene as S


def half(a):
    """texture / environment source images are made half-representable: the reference stores MIP level 0 in half
    precision, so the plugin's level 0 is then exactly the image handed over"""
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


def _sky(w, h):
    y, x = np.mgrid[0:h, 0:w]
    el = (0.5 - (y + 0.5) / h) * np.pi
    az = (x + 0.5) / w * 2 * np.pi
    base = np.stack([0.3 + 0.5 * np.clip(np.sin(el), 0, 1), 0.4 + 0.4 * np.clip(np.sin(el), 0, 1), 0.6 + 0.3 * np.cos(az)], -1)
    sun = np.exp(-(((az - 1.0) * 2) ** 2 + ((el - 0.6) * 4) ** 2))[..., None] * np.array([30.0, 25.0, 18.0])
    return (base + sun).astype(np.float32)


def _rot(axis, deg):
    a = np.asarray(axis, np.float64); a /= np.linalg.norm(a)
    t = np.radians(deg); c, s = np.cos(t), np.sin(t)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    R = np.eye(4); R[:3, :3] = c * np.eye(3) + s * K + (1 - c) * np.outer(a, a)
    return R.astype(np.float32)


def _sphere_uvs(N):
    N = np.asarray(N, np.float64)
    u = (np.arctan2(N[:, 2], N[:, 0]) / (2 * np.pi)) % 1.0
    v = np.arccos(np.clip(N[:, 1], -1, 1)) / np.pi
    return np.stack([u, v], -1).astype(np.float32)


def _checker(n, cell):
    y, x = np.mgrid[0:n, 0:n]
    c = ((x // cell + y // cell) % 2).astype(np.float32)
    return np.stack([0.1 + 0.8 * c, 0.15 + 0.7 * c, 0.2 + 0.5 * (1 - c)], -1).astype(np.float32)


def cornell(gauss, mip, res=(24, 24)):
    return S.cornell_box(res[0], res[1], gauss)


def zoo(gauss, mip):
    sb = S.cornell_box(32, 32, gauss)
    cu = dict(eta=S.CU_ETA, k=S.CU_K)
    mats = [sb.twosided(sb.roughconductor(alpha=0.2, distribution="ggx", **cu)),
            sb.twosided(sb.roughconductor(alpha=0.05, alpha_v=0.3, **cu)),
            sb.twosided(sb.roughconductor(alpha=0.15, sample_visible=False, **cu), sb.diffuse((0.2, 0.7, 0.3))),
            sb.roughconductor(alpha=0.3, **cu),
            sb.dielectric(1.33, 1.0)]
    for i, m in enumerate(mats):
        P, T, N = S.sphere_mesh((90 + 95 * i, 420 - 60 * (i % 2), 150 + 60 * i), 45.0, 24, 12)
        sb.mesh(P, T, m, normals=N, uvs=_sphere_uvs(N))       # anisotropic BSDFs need texture coordinates (trimesh.cpp:683-690)
    return sb


def glass(gauss, mip):
    return S.glass_room(32, 18, gauss, detail=0.3)


def atrium(gauss, mip):
    return S.atrium(32, 18, gauss, detail=0.3)


def _env_scene(gauss, env, res=(40, 24)):
    sb = S.SceneBuilder()
    env(sb)                                                    # environment emitters first: the order of Scene::getEmitters()
    floor = sb.diffuse((0.4, 0.45, 0.5))
    mats = [sb.diffuse((0.7, 0.3, 0.2)), sb.twosided(sb.diffuse((0.2, 0.6, 0.3))),
            sb.roughconductor(alpha=0.2, eta=S.CU_ETA, k=S.CU_K), sb.dielectric(1.5, 1.0)]
    sb.quad((-6, 0, -6), (6, 0, -6), (6, 0, 6), (-6, 0, 6), floor, facing=(0, 1, 0))
    for i, m in enumerate(mats):
        P, T, N = S.sphere_mesh((-3 + 2 * i, 0.8, 0.5 * (i % 2)), 0.8, 16, 8)
        sb.mesh(P, T, m, normals=N)
    sb.quad((-1, 4, -1), (1, 4, -1), (1, 4, 1), (-1, 4, 1), sb.diffuse((0, 0, 0)), facing=(0, -1, 0), radiance=(8, 8, 6))
    sb.perspective((0, 3, -9), (0, 0.5, 0), (0, 1, 0), 40.0)
    sb.hdrfilm(res[0], res[1], gauss)
    return sb


def const_env(gauss, mip):
    return _env_scene(gauss, lambda sb: sb.constant((0.9, 1.0, 1.2), sampling_weight=0.7))


def envmap(gauss, mip):
    rng = np.random.default_rng(5)
    tex = half(_sky(64, 32) * rng.uniform(0.5, 1.5, (32, 64, 1)))
    levels = mip("envmap", tex, kind="envmap")
    return _env_scene(gauss, lambda sb: sb.envmap(levels[0], scale=0.8, to_world=_rot((1, 0.3, 0.2), 70.0), pyramid=levels))


def textures(gauss, mip):
    rng = np.random.default_rng(11)
    noise = half(rng.uniform(0.05, 0.95, (24, 40, 3)))
    chk = half(_checker(64, 8))
    sb = S.SceneBuilder()
    sb.constant((0.4, 0.5, 0.7))

    def bm(key, img, **kw):
        lv = mip(key, img, kind="texture", wrap_u=kw.get("wrap", "repeat"), wrap_v=kw.get("wrap_v"),
                 filter_type=kw.get("filter_type", "ewa"), max_anisotropy=kw.get("max_anisotropy", 20.0))
        return sb.bitmap(lv[0], pyramid=lv, **kw)
    t_floor = bm("floor", chk, filter_type="ewa", uscale=6.0, vscale=6.0)
    t_tri = bm("tri", noise, filter_type="trilinear", wrap="mirror", wrap_v="clamp", uscale=2.0, uoffset=0.3)
    t_bil = bm("bil", noise, filter_type="bilinear", wrap="zero", wrap_v="one", uscale=1.5, vscale=1.5, voffset=-0.2)
    t_near = bm("near", chk[:32, :24], filter_type="nearest")
    t_ewa2 = bm("ewa2", noise[:15, :25], filter_type="ewa", max_anisotropy=2.0, uscale=3.0, wrap="clamp")
    sb.quad((-8, 0, -8), (8, 0, -8), (8, 0, 8), (-8, 0, 8), sb.diffuse(texture=t_floor), facing=(0, 1, 0), uvs=True)
    mats = [sb.diffuse(texture=t_tri), sb.twosided(sb.diffuse(texture=t_bil)), sb.twosided(sb.diffuse(texture=t_near), sb.diffuse((0.3, 0.3, 0.3))),
            sb.diffuse(texture=t_ewa2), sb.roughconductor(alpha=0.05, alpha_v=0.3, eta=S.CU_ETA, k=S.CU_K, texture=t_tri),   # textured specularReflectance
            sb.dielectric(1.5, 1.0, texture=t_floor)]
    for i, m in enumerate(mats):
        P, T, N = S.sphere_mesh((-5 + 2 * i, 0.8, 0.4 * (i % 2)), 0.8, 16, 8)
        sb.mesh(P, T, m, normals=N if i % 2 == 0 else None, uvs=_sphere_uvs(N))
    sb.quad((-2, 5, -2), (2, 5, -2), (2, 5, 2), (-2, 5, 2), sb.diffuse((0, 0, 0)), facing=(0, -1, 0), radiance=(10, 10, 9))
    sb.perspective((0, 3.5, -11), (0, 0.4, 0), (0, 1, 0), 42.0)
    sb.hdrfilm(48, 32, gauss)
    return sb


def roughness_maps(gauss, mip):
    """`bitmap` textures on the roughness of roughconductor (child "alpha": one texture for both axes; "alphaU" + "alphaV": two
    textures, anisotropic) and on the dielectric's specularTransmittance (roughconductor.cpp:275-280,424-431; dielectric.cpp:307,363)"""
    rng = np.random.default_rng(23)
    a_chk = half(0.04 + 0.5 * _checker(64, 8))                                  # roughness 0.04 / 0.54 in a checkerboard
    a_noise = half(rng.uniform(0.03, 0.6, (20, 32, 3)))
    a_noise2 = half(rng.uniform(0.05, 0.4, (16, 16, 3)))
    tint = half(rng.uniform(0.3, 1.0, (12, 20, 3)))
    refl = half(rng.uniform(0.4, 1.0, (16, 24, 3)))
    sb = S.SceneBuilder()
    sb.constant((0.5, 0.55, 0.7))

    def bm(key, img, **kw):
        lv = mip(key, img, kind="texture", wrap_u=kw.get("wrap", "repeat"), wrap_v=kw.get("wrap_v"),
                 filter_type=kw.get("filter_type", "ewa"), max_anisotropy=kw.get("max_anisotropy", 20.0))
        return sb.bitmap(lv[0], pyramid=lv, **kw)
    t_floor = bm("rm_floor", a_chk, filter_type="ewa", uscale=5.0, vscale=5.0)
    t_u = bm("rm_u", a_noise, filter_type="trilinear", wrap="mirror", uscale=2.0)
    t_v = bm("rm_v", a_noise2, filter_type="bilinear", vscale=3.0, voffset=0.1)
    t_n = bm("rm_n", a_noise2[:9, :13], filter_type="nearest", uscale=4.0, vscale=2.0)
    t_tint = bm("rm_tint", tint, filter_type="ewa", uscale=2.0, vscale=2.0)
    t_refl = bm("rm_refl", refl, filter_type="ewa", max_anisotropy=4.0)
    sb.quad((-8, 0, -8), (8, 0, -8), (8, 0, 8), (-8, 0, 8), sb.twosided(sb.roughconductor(eta=S.CU_ETA, k=S.CU_K, alpha_texture=t_floor)), facing=(0, 1, 0), uvs=True)
    mats = [sb.roughconductor(eta=S.CU_ETA, k=S.CU_K, alpha_texture=t_u, alpha_v_texture=t_v),                       # anisotropic: two textures
            sb.twosided(sb.roughconductor(eta=(0.2, 0.9, 1.1), k=(3.9, 2.4, 2.2), distribution="ggx", alpha_texture=t_n, texture=t_refl)),
            sb.dielectric(1.5, 1.0, transmittance_texture=t_tint),
            sb.roughconductor(eta=S.CU_ETA, k=S.CU_K, alpha_texture=t_v, sample_visible=False),
            sb.dielectric(1.33, 1.0, specular_transmittance=(0.9, 0.95, 1.0), texture=t_refl, transmittance_texture=t_tint),
            sb.diffuse((0.6, 0.5, 0.4))]
    for i, m in enumerate(mats):
        P, T, N = S.sphere_mesh((-5 + 2 * i, 0.8, 0.4 * (i % 2)), 0.8, 16, 8)
        sb.mesh(P, T, m, normals=N if i % 2 == 0 else None, uvs=_sphere_uvs(N))
    sb.quad((-2, 5, -2), (2, 5, -2), (2, 5, 2), (-2, 5, 2), sb.diffuse((0, 0, 0)), facing=(0, -1, 0), radiance=(10, 10, 9))
    sb.perspective((0, 3.5, -11), (0, 0.4, 0), (0, 1, 0), 42.0)
    sb.hdrfilm(48, 32, gauss)
    return sb


def box_mip(key, image, kind, **kw):
    """stand-in for the reference-built pyramids where the reference is not needed (GPU-vs-oracle fuzz): box-filtered levels"""
    return S.mip_pyramid(np.ascontiguousarray(image, np.float32))


def random_scene(gauss, seed, res=(24, 16), mip=None):
    """fuzz input: random spheres (smooth / flat, with texture coordinates) and a triangle soup, 2-5 random materials of every
    kind, 1-2 quad lights, sometimes a constant environment, random camera.  Returns (SceneBuilder, render parameters)."""
    rng = np.random.default_rng(seed)
    sb = S.SceneBuilder()
    env = rng.random()
    if mip is not None and env < 0.25:      # envmap (rotated), with its MIP pyramid: directly visible pixels are EWA-filtered
        w = int(rng.integers(8, 40)); h = int(rng.integers(4, 24))
        lv = mip("env%d" % seed, half(rng.uniform(0.0, 2.0, (h, w, 3)) * rng.uniform(0.2, 3.0)), kind="envmap")
        sb.envmap(lv[0], scale=float(rng.uniform(0.3, 2.0)), to_world=_rot(rng.normal(size=3), float(rng.uniform(0, 360))),
                  sampling_weight=float(rng.uniform(0.3, 2)), pyramid=lv)
    elif env < 0.5:
        sb.constant(tuple(rng.uniform(0.1, 1.0, 3)), sampling_weight=float(rng.uniform(0.3, 2)))
    mats = []
    if mip is not None:                     # bitmap reflectance textures with random lookup parameters
        for i in range(rng.integers(1, 4)):
            w = int(rng.integers(3, 50)); h = int(rng.integers(3, 50))
            wraps = ["repeat", "clamp", "mirror", "zero", "one"]
            kw = dict(wrap=wraps[rng.integers(5)], wrap_v=wraps[rng.integers(5)], filter_type=["nearest", "bilinear", "trilinear", "ewa"][rng.integers(4)],
                      max_anisotropy=float(rng.uniform(1.0, 20.0)))
            lv = mip("tex%d_%d" % (seed, i), half(rng.uniform(0.0, 1.0, (h, w, 3))), kind="texture", wrap_u=kw["wrap"], wrap_v=kw["wrap_v"],
                     filter_type=kw["filter_type"], max_anisotropy=kw["max_anisotropy"])
            t = sb.bitmap(lv[0], pyramid=lv, uscale=float(rng.uniform(0.3, 8)), vscale=float(rng.uniform(0.3, 8)),
                          uoffset=float(rng.uniform(-1, 1)), voffset=float(rng.uniform(-1, 1)), **kw)
            kind = rng.integers(0, 7)
            if kind == 2:
                d = sb.roughconductor(alpha=float(rng.uniform(0.05, 0.5)), eta=S.CU_ETA, k=S.CU_K, texture=t)      # specularReflectance
            elif kind == 3:
                mats.append(sb.dielectric(float(rng.uniform(1.2, 1.8)), 1.0, texture=t)); continue
            elif kind == 4:                                  # the same image as the roughness of both axes (child "alpha")
                d = sb.roughconductor(eta=S.CU_ETA, k=S.CU_K, alpha_texture=t, distribution=["beckmann", "ggx"][rng.integers(2)], sample_visible=bool(rng.integers(2)))
            elif kind == 5:                                  # "alphaU" textured, alphaV constant: anisotropic
                d = sb.roughconductor(eta=S.CU_ETA, k=S.CU_K, alpha_texture=t, alpha_v=float(rng.uniform(0.05, 0.4)))
                sb.materials[d].alpha_v_texture = 0
            elif kind == 6:
                mats.append(sb.dielectric(float(rng.uniform(1.2, 1.8)), 1.0, transmittance_texture=t)); continue
            else:
                d = sb.diffuse(texture=t)
            mats.append(d if rng.random() < 0.5 else sb.twosided(d))
    for i in range(rng.integers(1 if mats else 2, 6)):
        t = rng.integers(0, 5)
        if t == 0:
            m = sb.diffuse(tuple(rng.uniform(0, 0.9, 3)))
        elif t == 1:
            m = sb.dielectric(float(rng.uniform(1.1, 2.0)), 1.0)
        elif t == 2:
            m = sb.roughconductor(alpha=float(rng.uniform(0.02, 0.6)), distribution=["beckmann", "ggx"][rng.integers(2)],
                                  sample_visible=bool(rng.integers(2)), eta=S.CU_ETA, k=S.CU_K)
        elif t == 3:
            m = sb.twosided(sb.diffuse(tuple(rng.uniform(0, 0.9, 3))))
        else:
            m = sb.twosided(sb.roughconductor(alpha=float(rng.uniform(0.05, 0.4)), alpha_v=float(rng.uniform(0.05, 0.4)), eta=S.CU_ETA, k=S.CU_K),
                            sb.diffuse(tuple(rng.uniform(0, 0.9, 3))))
        mats.append(m)
    for i in range(rng.integers(2, 6)):
        P, T, N = S.sphere_mesh(tuple(rng.uniform(-2, 2, 3)), float(rng.uniform(0.3, 1.0)), 8, 5)
        sb.mesh(P, T, mats[rng.integers(len(mats))], normals=N if rng.random() < 0.6 else None, uvs=_sphere_uvs(N))
    n = rng.integers(5, 40)
    c = rng.uniform(-3, 3, (n, 1, 3))
    P = (c + rng.normal(scale=0.8, size=(n, 3, 3))).astype(np.float32).reshape(-1, 3)
    sb.mesh(P, np.arange(3 * n, dtype=np.uint32).reshape(n, 3), sb.diffuse(tuple(rng.uniform(0.1, 0.9, 3))))
    for i in range(rng.integers(1, 3)):
        o = rng.uniform(-2, 2, 3); o[1] = rng.uniform(2.5, 4)
        sb.quad(tuple(o + [-0.5, 0, -0.5]), tuple(o + [0.5, 0, -0.5]), tuple(o + [0.5, 0, 0.5]), tuple(o + [-0.5, 0, 0.5]),
                sb.diffuse((0, 0, 0)), facing=(0, -1, 0), radiance=tuple(rng.uniform(2, 20, 3)))
    eye = rng.uniform(-1, 1, 3) * [4, 1, 4]; eye[2] = -6
    sb.perspective(tuple(eye), (0, 0, 0), (0, 1, 0), float(rng.uniform(30, 80)))
    if res == "random":                                 # ragged film sizes and crop windows
        frng = np.random.default_rng(5000 + seed)
        w, h = int(frng.integers(5, 90)), int(frng.integers(5, 70))
        crop = None
        if frng.random() < 0.5:
            cw, ch = int(frng.integers(1, w + 1)), int(frng.integers(1, h + 1))
            crop = (int(frng.integers(0, w - cw + 1)), int(frng.integers(0, h - ch + 1)), cw, ch)
        sb.hdrfilm(w, h, gauss, crop=crop)
    else:
        sb.hdrfilm(res[0], res[1], gauss)
    prng = np.random.default_rng(1000 + seed)
    if seed % 2 == 0:
        kw = dict(spp=2, max_depth=int(prng.integers(2, 12)), rr_depth=int(prng.integers(1, 6)), strict_normals=int(prng.integers(2)),
                  hide_emitters=int(prng.integers(2)))
    else:
        kw = dict(spp=1, integrator=A.PHIP_INTEGRATOR_DIRECT, emitter_samples=int(prng.integers(0, 3)), bsdf_samples=int(prng.integers(1, 3)),
                  strict_normals=int(prng.integers(2)))
    return sb, kw


PATH, DIRECT, VOLPATH = A.PHIP_INTEGRATOR_PATH, A.PHIP_INTEGRATOR_DIRECT, A.PHIP_INTEGRATOR_VOLPATH_SIMPLE

# (case name, scene builder, render parameters)
CASES = [
    ("cornell_path", cornell, dict(spp=4, max_depth=-1)),
    ("cornell_rr2", cornell, dict(spp=4, max_depth=-1, rr_depth=2)),
    ("cornell_md2", cornell, dict(spp=2, max_depth=2)),
    ("cornell_hide_strict", cornell, dict(spp=2, max_depth=6, hide_emitters=1, strict_normals=1)),
    ("cornell_direct_1_1", cornell, dict(spp=2, integrator=DIRECT, emitter_samples=1, bsdf_samples=1)),
    ("cornell_direct_3_2", cornell, dict(spp=2, integrator=DIRECT, emitter_samples=3, bsdf_samples=2)),
    ("cornell_direct_0_2", cornell, dict(spp=2, integrator=DIRECT, emitter_samples=0, bsdf_samples=2)),
    ("cornell_direct_2_0", cornell, dict(spp=2, integrator=DIRECT, emitter_samples=2, bsdf_samples=0, hide_emitters=1)),
    ("zoo_path", zoo, dict(spp=2, max_depth=8)),
    ("zoo_strict", zoo, dict(spp=1, max_depth=4, strict_normals=1)),
    ("zoo_direct", zoo, dict(spp=1, integrator=DIRECT, emitter_samples=2, bsdf_samples=2)),
    ("glass_path", glass, dict(spp=2, max_depth=16)),
    ("atrium_path", atrium, dict(spp=2, max_depth=8)),
    ("const_env_path", const_env, dict(spp=2, max_depth=6)),
    ("const_env_hide", const_env, dict(spp=1, max_depth=4, hide_emitters=1, strict_normals=1)),
    ("const_env_direct", const_env, dict(spp=1, integrator=DIRECT, emitter_samples=2, bsdf_samples=2)),
    ("envmap_path", envmap, dict(spp=2, max_depth=6)),
    ("envmap_direct", envmap, dict(spp=1, integrator=DIRECT, emitter_samples=2, bsdf_samples=3)),
    ("textures_path", textures, dict(spp=2, max_depth=6)),
    ("textures_direct", textures, dict(spp=1, integrator=DIRECT, emitter_samples=2, bsdf_samples=2)),
    ("roughness_maps_path", roughness_maps, dict(spp=2, max_depth=8)),
    ("roughness_maps_direct", roughness_maps, dict(spp=1, integrator=DIRECT, emitter_samples=2, bsdf_samples=2)),
    # round 5: the sibling integrator `volpath_simple` on media-free scenes (src/integrators/path/volpath_simple.cpp)
    ("cornell_volpath", cornell, dict(spp=4, max_depth=-1, integrator=VOLPATH)),
    ("cornell_volpath_rr2", cornell, dict(spp=2, max_depth=-1, rr_depth=2, integrator=VOLPATH)),
    ("cornell_volpath_md1", cornell, dict(spp=2, max_depth=1, integrator=VOLPATH)),
    ("cornell_volpath_md2", cornell, dict(spp=2, max_depth=2, integrator=VOLPATH)),
    ("cornell_volpath_md3", cornell, dict(spp=2, max_depth=3, integrator=VOLPATH)),
    ("cornell_volpath_hide_strict", cornell, dict(spp=2, max_depth=6, hide_emitters=1, strict_normals=1, integrator=VOLPATH)),
    ("zoo_volpath", zoo, dict(spp=2, max_depth=8, integrator=VOLPATH)),
    ("zoo_volpath_strict", zoo, dict(spp=1, max_depth=4, strict_normals=1, integrator=VOLPATH)),
    ("glass_volpath", glass, dict(spp=2, max_depth=16, integrator=VOLPATH)),
    ("const_env_volpath", const_env, dict(spp=2, max_depth=6, rr_depth=2, integrator=VOLPATH)),
    ("const_env_volpath_hide", const_env, dict(spp=1, max_depth=4, hide_emitters=1, strict_normals=1, integrator=VOLPATH)),
    ("envmap_volpath", envmap, dict(spp=2, max_depth=6, integrator=VOLPATH)),
    ("textures_volpath", textures, dict(spp=2, max_depth=6, integrator=VOLPATH)),
]


def params(kw):
    return A.default_render_params(block_size=256, **kw)      # one image block: a single sampler stream, row-major pixels
