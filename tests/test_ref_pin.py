"""Pins the oracle to the REFERENCE ITSELF: /root/reference's own libcore + librender + plugins, compiled in place into
oracle/_ref by oracle/Makefile.ref (Boost / Eigen header-shimmed, nothing copied), driven through oracle/ref_driver.cpp.
The oracle's libm build with the reference's sampler stream (`independent`: SFMT19937, one clone) must reproduce the
reference's per-sample radiance and its ImageBlock accumulator BIT FOR BIT; function-level hooks pin the pieces.
Runs where the reference tree (or a prebuilt oracle/_ref) exists; tests/test_golden.py carries the same check everywhere
through the committed fixture tests/golden/ref_renders.npz."""
import os

import numpy as np
import pytest

import ref_scenes as RS
from mitsuba_amd import _abi as A, scene as S


@pytest.fixture(scope="module")
def ref():
    from oracle import ref_ffi
    if not ref_ffi.available():
        pytest.skip("neither /root/reference nor a prebuilt oracle/_ref is present")
    ref_ffi.build()
    ref_ffi.lib()
    return ref_ffi


@pytest.fixture(scope="module")
def olibm(oracle):
    oracle.build(libm=True)
    return oracle


def live_mip(ref):
    def mip(key, image, kind, wrap_u="repeat", wrap_v=None, filter_type="ewa", max_anisotropy=None):
        return ref.RefMip(image, kind=kind, wrap_u=wrap_u, wrap_v=wrap_v, filter_type=filter_type, max_anisotropy=max_anisotropy).levels
    return mip


@pytest.mark.parametrize("name,build,kw", RS.CASES, ids=[c[0] for c in RS.CASES])
def test_oracle_reproduces_the_reference_bit_for_bit(ref, olibm, name, build, kw):
    gauss = olibm.gaussian_filter(0.5, libm=True)
    desc = build(gauss, live_mip(ref)).desc()
    p = RS.params(kw)
    rs = ref.RefScene(desc)
    rfilm, rsmp = rs.render(p)
    osc = olibm.OracleScene(desc, libm=True)
    ofilm, osmp, _ = osc.render(p, threads=1, sampler="sfmt", want_samples=True)
    assert np.isfinite(rsmp).all() and rsmp[..., :3].max() > 0
    assert np.array_equal(rsmp.view(np.uint32), osmp.view(np.uint32)), \
        "%.4f%% of the samples are bit-identical" % (100 * (rsmp.view(np.uint32) == osmp.view(np.uint32)).all(-1).mean())
    assert np.array_equal(rfilm.view(np.uint32), ofilm.view(np.uint32))        # ImageBlock::put, same order of additions
    # SamplingIntegrator::renderBlock itself (no per-sample hook) gives the same block
    rfilm2, _ = rs.render(p, want_samples=False)
    assert np.array_equal(rfilm2.view(np.uint32), rfilm.view(np.uint32))
    rs.close(); osc.close()


@pytest.mark.parametrize("name,build,kw", RS.CASES, ids=[c[0] for c in RS.CASES])
def test_reference_on_the_parity_stream(ref, oracle, olibm, name, build, kw):
    """the counter-based parity stream itself, validated on the real integrators: the reference's `path` / `direct` fed by
    oracle/ref_glue/ctr_sampler.cpp (a Sampler plugin for the reference that serves pcg4d(pixel, sample, block, seed) purely
    by call order -- it knows nothing about the scene, dielectrics included) consume exactly the numbers the oracle's -- and hence the GPU's -- ctr renders
    consume: bit-identical with the libm build, and the parity build (= the GPU, bit for bit) is within the north-star
    tolerance of the REFERENCE ON THE SAME SAMPLES by orders of magnitude"""
    gauss = olibm.gaussian_filter(0.5, libm=True)
    desc = build(gauss, live_mip(ref)).desc()
    p = RS.params(kw)
    osc = olibm.OracleScene(desc, libm=True)
    rs = ref.RefScene(desc)
    rfilm, rsmp = rs.render(p, sampler="ctr")
    ofilm, osmp, _ = osc.render(p, threads=1, sampler="ctr", want_samples=True)
    assert np.array_equal(rsmp.view(np.uint32), osmp.view(np.uint32))
    assert np.array_equal(rfilm.view(np.uint32), ofilm.view(np.uint32))
    pfilm, psmp, _ = oracle.OracleScene(desc).render(p, threads=1, sampler="ctr", want_samples=True)
    rel = np.linalg.norm(psmp - rsmp) / np.linalg.norm(rsmp)
    assert rel < 1e-4, rel                                  # measured: 0 .. 1e-6
    rs.close()


def test_random_scenes_fuzz(ref, olibm):
    """random scenes (ref_scenes.random_scene: sphere and triangle soups, every material kind, bitmap textures with random
    size / filter / wrap modes / uv transform, rotated envmaps, random integrator parameters): still bit for bit.
    3000 seeds were run once during development (all identical); 24 here"""
    gauss = olibm.gaussian_filter(0.5, libm=True)
    for seed in range(24):
        sb, kw = RS.random_scene(gauss, seed, mip=live_mip(ref))
        desc = sb.desc()
        p = RS.params(kw)
        rs = ref.RefScene(desc); rfilm, rsmp = rs.render(p); rs.close()
        osc = olibm.OracleScene(desc, libm=True)
        ofilm, osmp, _ = osc.render(p, threads=1, sampler="sfmt", want_samples=True); osc.close()
        assert np.array_equal(rsmp.view(np.uint32), osmp.view(np.uint32)), (seed, kw)
        assert np.array_equal(rfilm.view(np.uint32), ofilm.view(np.uint32)), (seed, kw)


def test_parity_build_stays_within_tolerance_of_the_reference(ref, oracle, olibm):
    """the parity oracle (phip_fmath.h transcendentals -- what the GPU is compared with) against the reference on the
    reference's own sample stream: identical control flow, differences of a few ulp in a minority of the samples"""
    gauss = olibm.gaussian_filter(0.5, libm=True)
    for name, build, kw in RS.CASES:
        if name not in ("cornell_path", "zoo_path", "glass_path", "envmap_path", "textures_path"):
            continue
        desc = build(gauss, live_mip(ref)).desc()
        p = RS.params(kw)
        _, rsmp = ref.RefScene(desc).render(p)
        _, osmp, _ = oracle.OracleScene(desc).render(p, threads=1, sampler="sfmt", want_samples=True)
        rel = np.linalg.norm(rsmp - osmp) / np.linalg.norm(rsmp)
        same = (rsmp.view(np.uint32) == osmp.view(np.uint32)).all(-1).mean()
        print("%s: phip_fmath oracle vs reference: identical %.4f rel L2 %.2e" % (name, same, rel))
        assert rel < 1e-3                                       # north_star tolerance; in practice ~1e-7


def test_function_level_hooks_match(ref, olibm):
    """Scene::rayIntersect, Sensor::sampleRayDifferential, BSDF::sample / eval / pdf, Scene::sampleEmitterDirect +
    pdfEmitterDirect: the reference's functions against the oracle's restatements on random inputs"""
    import ctypes as C
    gauss = olibm.gaussian_filter(0.5, libm=True)
    rng = np.random.default_rng(7)
    desc = RS.zoo(gauss, None).desc()
    rs = ref.RefScene(desc); osc = olibm.OracleScene(desc, libm=True)
    # rays
    n = 4000
    o = rng.uniform(50, 500, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), np.float32); rays[:, :3] = o; rays[:, 3] = 1e-4; rays[:, 4:7] = d; rays[:, 7] = np.inf
    rh = rs.trace(rays)
    oh, _, _ = osc.trace(rays)
    hit = np.isfinite(rh[:, 0])
    assert hit.mean() > 0.8
    assert np.array_equal(hit, oh.view(np.uint32)[:, 3] != 0xFFFFFFFF)
    assert np.array_equal(rh[hit, 0].view(np.uint32), oh[hit, 0].view(np.uint32))     # t
    # camera rays incl. differentials
    for sx, sy in rng.uniform(0, 32, (50, 2)):
        r = rs.camera_ray(float(sx), float(sy)); oc = osc.camera_ray(float(sx), float(sy))
        assert np.array_equal(r[:3], oc[:3]) and np.array_equal(r[3:6], oc[4:7]) and r[6] == oc[3] and r[7] == oc[7]
    # BSDFs: every material of the scene
    L = olibm.lib(libm=True)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    m = 3000
    for mat in range(desc.n_materials):
        wi = rng.normal(size=(m, 3)).astype(np.float32); wi /= np.linalg.norm(wi, axis=1, keepdims=True)
        smp = np.minimum(rng.random((m, 2)).astype(np.float32), np.float32(1 - 2 ** -24))
        rwo, rw, rpdf, rdelta = rs.bsdf_sample(mat, wi, smp)
        owo = np.zeros((m, 3), np.float32); ow = np.zeros((m, 3), np.float32); opdf = np.zeros(m, np.float32); odelta = np.zeros(m, np.uint8)
        L.oracle_bsdf_sample(osc.h, mat, m, fp(wi), fp(smp), fp(owo), fp(ow), fp(opdf), odelta.ctypes.data_as(C.POINTER(C.c_uint8)))
        assert np.array_equal(rw.view(np.uint32), ow.view(np.uint32)), mat
        assert np.array_equal(rpdf.view(np.uint32), opdf.view(np.uint32)), mat
        ok = rpdf > 0
        assert np.array_equal(rwo[ok].view(np.uint32), owo[ok].view(np.uint32)), mat
        wo = rng.normal(size=(m, 3)).astype(np.float32); wo /= np.linalg.norm(wo, axis=1, keepdims=True)
        rv, rp = rs.bsdf_eval_pdf(mat, wi, wo)
        ov = np.zeros((m, 3), np.float32); op = np.zeros(m, np.float32)
        L.oracle_bsdf_eval_pdf(osc.h, mat, m, fp(wi), fp(wo), fp(ov), fp(op))
        assert np.array_equal(rv.view(np.uint32), ov.view(np.uint32)), mat
        assert np.array_equal(rp.view(np.uint32), op.view(np.uint32)), mat
    # emitter sampling from points inside the box
    for _ in range(20):
        refp = rng.uniform(100, 450, 3).astype(np.float32)
        refn = rng.normal(size=3).astype(np.float32); refn /= np.linalg.norm(refn)
        smp = rng.random((200, 2)).astype(np.float32)
        rd, rdist, rpdf, rval, rchk = rs.sample_emitter(refp, refn, smp)
        od = np.zeros((200, 3), np.float32); odist = np.zeros(200, np.float32); opdf = np.zeros(200, np.float32)
        oval = np.zeros((200, 3), np.float32); ochk = np.zeros(200, np.float32)
        L.oracle_sample_emitter(osc.h, fp(refp), fp(refn), 200, fp(smp), fp(od), fp(odist), fp(opdf), fp(oval), fp(ochk))
        ok = rpdf > 0
        assert np.array_equal(ok, opdf > 0)
        assert np.array_equal(rval.view(np.uint32), oval.view(np.uint32))
        assert np.array_equal(rpdf[ok].view(np.uint32), opdf[ok].view(np.uint32))
        assert np.array_equal(rd[ok].view(np.uint32), od[ok].view(np.uint32))
        assert np.array_equal(rchk[ok].view(np.uint32), ochk[ok].view(np.uint32))


def test_reference_rejects_anisotropic_bsdf_without_texcoords(ref, olibm):
    """TriMesh::computeUVTangents (trimesh.cpp:683-690) is an error for an anisotropic BSDF on a mesh without texture
    coordinates; the oracle (and libphip) refuse the same scene"""
    gauss = olibm.gaussian_filter(0.5, libm=True)
    sb = S.cornell_box(8, 8, gauss)
    P, T, N = S.sphere_mesh((200, 200, 200), 50.0, 8, 4)
    sb.mesh(P, T, sb.roughconductor(alpha=0.05, alpha_v=0.3, eta=S.CU_ETA, k=S.CU_K), normals=N)
    with pytest.raises(RuntimeError, match="texture coordinates"):
        ref.RefScene(sb.desc())
    with pytest.raises(RuntimeError, match="texture coordinates"):
        olibm.OracleScene(sb.desc(), libm=True)


def test_reference_render_job_multi_threaded(ref, oracle):
    """the reference's complete render (RenderJob -> BlockedRenderProcess on the Scheduler's LocalWorkers, Hilbert-curve
    pixel order, one sampler clone per worker -- bench.py's CPU baseline) agrees with the oracle's parity-stream render
    statistically, and repeated jobs on one scene work"""
    gauss = oracle.gaussian_filter(0.5)
    desc = S.cornell_box(96, 96, gauss).desc()
    rs = ref.RefScene(desc)
    p = A.default_render_params(spp=64, max_depth=6)
    rgb, sec = rs.render_job(p, threads=4)
    assert sec > 0 and np.isfinite(rgb).all()
    film, _, _ = oracle.OracleScene(desc).render(p)
    o = oracle.develop(film)
    assert abs(rgb.mean() - o.mean()) / o.mean() < 0.02
    assert np.linalg.norm(rgb - o) / np.linalg.norm(o) < 0.15          # two independent 64-spp renders
    rgb2, _ = rs.render_job(A.default_render_params(spp=8, integrator=A.PHIP_INTEGRATOR_DIRECT, emitter_samples=2, bsdf_samples=2), threads=4)
    film2, _, _ = oracle.OracleScene(desc).render(A.default_render_params(spp=8, integrator=A.PHIP_INTEGRATOR_DIRECT, emitter_samples=2, bsdf_samples=2))
    o2 = oracle.develop(film2)
    assert abs(rgb2.mean() - o2.mean()) / o2.mean() < 0.03


def test_reference_with_correctly_rounded_libm_is_bit_identical_to_the_parity_build():
    """The one difference between Mitsuba 0.6 and the arithmetic the GPU kernels run (oracle parity build) is libm: with
    oracle/_build/libcrm.so LD_PRELOADed -- the reference's sincosf / expf / logf / acosf / atan2f / atanf / tanf / powf calls
    answered by the correctly rounded functions of include/phip_fmath.h -- the REFERENCE's own `path` integrator produces the
    same bits, sample by sample, on the Cornell box, the atrium and the glass room (tools/ref_with_cr_libm.py; with glibc:
    97.7-99.1 %).  Hence the full-size image differences against the stock reference are libm rounding."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["make", "-C", os.path.join(root, "oracle"), "crm"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, LD_PRELOAD=os.path.join(root, "oracle", "_build", "libcrm.so"))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "ref_with_cr_libm.py")], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert {x["scene"] for x in rows} == {"cornell", "atrium", "glass_room"}
    for x in rows:
        assert "libcrm" in x["preload"]
        assert x["differing_samples"] == 0 and x["film_rel_l2"] == 0.0, x


def test_kd_tree_loses_a_silhouette_edge_hit_that_a_bvh_finds(ref, oracle, phip):
    """The one structural difference between the reference and path_hip, found by rendering BASELINE's C2 at full size
    (tests/test_gpu_parity.py::test_c2_at_full_size_against_the_oracle): camera sample 110 of pixel (602, 127) of the 1024x1024
    Cornell box passes through the edge x = 213 of the area light.  Every triangle test (TriAccel, bit-identical in all four
    implementations) accepts the hit on the light (v == 0 exactly), but x = 213 is also a split plane of the SAH kd-tree and the
    Wald distance comes out beyond the leaf's exit distance (sahkdtree3.h:263-272 intersects within [searchStart, searchEnd]),
    so Mitsuba's kd-tree -- and the oracle's restatement of it, bit for bit -- reports the ceiling behind it.  A BVH has no cell
    boundaries to lose a hit at: path_hip returns the structure-independent answer (the sweep over every triangle).  About five
    of C2's 268 M camera samples are affected (image rel. L2 3e-5)."""
    import ctypes as C
    gauss = oracle.gaussian_filter(0.5)
    desc = S.cornell_box(1024, 1024, gauss).desc()
    osc = oracle.OracleScene(desc)
    rs = ref.RefScene(desc)
    jit = np.zeros(4, np.float32)
    oracle.lib().oracle_ctr_block(127 * 1024 + 602, 110, 0, 0, jit.ctypes.data_as(C.POINTER(C.c_float)))
    ray = osc.camera_ray(np.float32(602) + jit[0], np.float32(127) + jit[1])
    assert np.array_equal(ray[[0, 1, 2, 4, 5, 6]], rs.camera_ray(np.float32(602) + jit[0], np.float32(127) + jit[1])[[0, 1, 2, 3, 4, 5]])
    rays = ray[None]
    kd, _, _ = osc.trace(rays, True, False)
    rt = rs.trace(rays)[0]
    assert rt[0] == kd[0, 0] and int(kd[0].view(np.uint32)[3]) == 3 and int(rt[3]) == 1      # Mitsuba and the oracle: the ceiling (shape 1, triangle 3)
    bf, _, _ = osc.trace(rays, True, False, bruteforce=True)
    assert int(bf[0].view(np.uint32)[3]) == 11 and bf[0, 2] == 0.0 and bf[0, 0] < kd[0, 0]   # every TriAccel: the light, on its edge (v == 0), in front
    P = np.ctypeslib.as_array(desc.positions, shape=(desc.n_vertices, 3)).copy()
    T = np.ctypeslib.as_array(desc.indices, shape=(desc.n_triangles, 3)).copy()
    hit = np.zeros((1, 4), np.float32)
    fpp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    rc = phip.phip_debug_host_trace_wide(fpp(P), len(P), T.ctypes.data_as(C.POINTER(C.c_uint32)), len(T), rays.ctypes.data_as(C.POINTER(A.phip_ray)), 1,
                                         hit.ctypes.data_as(C.POINTER(A.phip_hit)), 0, None, None, 0)
    assert rc == 0
    assert np.array_equal(hit.view(np.uint32), bf.view(np.uint32))                            # the BVH the GPU kernels walk: the sweep's answer, bit for bit
    hitp = ray[:3] + ray[4:7] * bf[0, 0]
    assert abs(hitp[0] - 213.0) < 1e-3 and abs(hitp[1] - 548.7) < 1e-3
    rs.close(); osc.close()


def test_ld_sampler_on_the_reference(ref, olibm):
    """PHIP_SAMPLER_LD, validated on the real integrator: Mitsuba's own `path` fed by oracle/ref_glue/ctr_sampler.cpp in its `ld` mode
    (the points are made by the reference's own qmc.h functions, served purely by call order: 2D requests 0..3 and 1D requests 0..3 of
    a sample are stratified, later ones fall back to the counter stream, exactly how ldsampler.cpp:212-226 hands them out) consumes the
    numbers the oracle -- and hence the GPU -- consumes: per-sample radiance bit-identical with the libm build"""
    import ref_scenes as RS
    gauss = olibm.gaussian_filter(0.5, libm=True)
    for name, desc, md in (("cornell", S.cornell_box(24, 24, gauss).desc(), -1), ("glass", S.glass_room(32, 18, gauss, detail=0.1).desc(), 12),
                           ("zoo", RS.zoo(gauss, None).desc(), 8)):
        for rr, spp in ((5, 8), (2, 4), (1, 16)):
            p = A.default_render_params(spp=spp, max_depth=md, rr_depth=rr, sampler=A.PHIP_SAMPLER_LD, block_size=256, seed=rr)
            osc = olibm.OracleScene(desc, libm=True)
            _, osmp, _ = osc.render(p, threads=1, want_samples=True)
            rs = ref.RefScene(desc)
            _, rsmp = rs.render(p, sampler="ctr")
            assert np.array_equal(osmp.view(np.uint32), rsmp.view(np.uint32)), (name, rr)
            p.sampler = A.PHIP_SAMPLER_CTR
            _, csmp, _ = osc.render(p, threads=1, want_samples=True)
            assert not np.array_equal(osmp, csmp)
            rs.close(); osc.close()
        # `direct`: sample arrays (more than one shading sample of a kind) and single samples in every combination
        for e, b, spp in ((1, 1, 8), (3, 2, 4), (0, 2, 4), (2, 0, 8), (1, 5, 2), (4, 1, 16)):
            p = A.default_render_params(spp=spp, sampler=A.PHIP_SAMPLER_LD, block_size=256, integrator=A.PHIP_INTEGRATOR_DIRECT, emitter_samples=e, bsdf_samples=b)
            osc = olibm.OracleScene(desc, libm=True)
            _, osmp, _ = osc.render(p, threads=1, want_samples=True)
            rs = ref.RefScene(desc)
            _, rsmp = rs.render(p, sampler="ctr")
            assert np.array_equal(osmp.view(np.uint32), rsmp.view(np.uint32)), (name, "direct", e, b)
            rs.close(); osc.close()


def test_ld_stream_converges_like_the_reference_ldsampler(ref, oracle):
    """PHIP_SAMPLER_LD is the construction of Mitsuba's ldsampler with other scrambles, so its numbers cannot be compared -- its
    quality can: on the Cornell box at 16 spp (path, maxDepth 3) the mean absolute error against a converged image is that of the
    reference's own `path` + `ldsampler` (within +-25 %; measured 0.0044 vs 0.0049), and both are well below the `independent`
    sampler's (0.0076).  (The mean SQUARED error is dominated by a few pixels and varies by a factor of two between runs of the reference.)"""
    gauss = oracle.gaussian_filter(0.5)
    desc = S.cornell_box(48, 48, gauss).desc()
    rs = ref.RefScene(desc)
    conv, _ = rs.render_job(A.default_render_params(spp=16384, max_depth=3), sampler="independent")
    mae = {}
    p = A.default_render_params(spp=16, max_depth=3)
    for smp in ("independent", "ldsampler"):
        mae[smp] = float(np.mean([np.mean(np.abs(rs.render_job(p, sampler=smp, threads=t)[0] - conv)) for t in (1, 2, 3)]))
    osc = oracle.OracleScene(desc)
    mae["phip_ld"] = float(np.mean([np.mean(np.abs(oracle.develop(osc.render(A.default_render_params(spp=16, max_depth=3, sampler=A.PHIP_SAMPLER_LD, seed=seed))[0]) - conv))
                                    for seed in range(3)]))
    print(mae)
    assert mae["ldsampler"] < 0.8 * mae["independent"] and mae["phip_ld"] < 0.8 * mae["independent"], mae
    assert 0.75 < mae["phip_ld"] / mae["ldsampler"] < 1.25, mae
    rs.close(); osc.close()


# ---- the reference's own XML loader and command-line front end -------------------------------------------------
def _mitsuba_cli():
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "mitsuba")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/mitsuba is not built")
    return exe


def test_scene_xml_through_the_reference_cli_equals_the_hand_assembled_scene(ref, gauss, tmp_path):
    """`mitsuba scene.xml`: the reference's OWN SceneHandler (src/librender/scenehandler.cpp, on the SAX sliver of oracle/ref_shims/xercesc)
    parses a scene file, its OWN front end (src/mitsuba/mitsuba.cpp: RenderQueue, RenderJob, Scheduler with two workers) renders it
    and HDRFilm writes a PFM -- bit for bit (Cornell box; the material zoo to 1e-5: its vertex normals are renormalised by the OBJ loader) the image
    ref_driver.cpp gets from the same description assembled by hand through
    PluginManager::createObject / addChild / configure (what every other pin test uses).  The parity sampler (ref_glue/ctr_sampler.cpp,
    an ordinary <sampler> plugin for the loader) makes the render independent of the worker schedule."""
    import subprocess
    import xml_scene as X
    exe = _mitsuba_cli()
    for name, build, md, spp, fmt in (("cornell", lambda: S.cornell_box(40, 32, gauss), 5, 4, "obj"), ("zoo", lambda: RS.zoo(gauss, None), 6, 2, "obj"),
                                      # meshes as .serialized files (Mitsuba's own format: written by TriMesh::serialize, read by the `serialized` plugin)
                                      ("cornell_serialized", lambda: S.cornell_box(40, 32, gauss), 5, 4, "serialized"),
                                      ("zoo_serialized", lambda: RS.zoo(gauss, None), 6, 2, "serialized")):
        desc = build().desc()
        out = tmp_path / name
        xml = X.write_scene_xml(desc, str(out), integrator="path", integrator_props=dict(maxDepth=md, rrDepth=5), sampler="ctr", spp=spp,
                                sampler_props=dict(seed=0, cropWidth=int(desc.film.crop_width), mode="path", rrDepth=5, sampleTotal=spp), mesh_format=fmt)
        r = subprocess.run([exe, "-q", "-p", "2", "-o", str(out / "cli.pfm"), xml], capture_output=True, text=True, cwd=str(out), timeout=600)
        assert r.returncode == 0 and os.path.exists(out / "cli.pfm"), r.stdout[-2000:] + r.stderr[-2000:]
        cli = X.read_pfm(str(out / "cli.pfm"))
        rs = ref.RefScene(desc)
        img, _ = rs.render_job(A.default_render_params(spp=spp, max_depth=md), threads=2, sampler="ctr")
        rs.close()
        assert cli.shape == img.shape
        assert np.isfinite(cli).all() and cli.mean() > 0.01
        if name != "zoo":           # (the serialized format stores the vertex normals as they are: bit-identical there too)
            assert (cli.view(np.uint32) == img.view(np.uint32)).all(), (name, float(np.abs(cli - img).max()))
        else:       # (vertex normals and uv coordinates pass through the OBJ loader, which renormalises: last-bit differences in the shading frames)
            assert np.abs(cli - img).max() <= 1e-5 * max(1.0, float(img.max())), (name, float(np.abs(cli - img).max()))


def test_scene_xml_errors_reach_the_reference_loader(ref, gauss, tmp_path):
    """malformed XML and an unknown plugin are reported by the loader (non-zero exit status, no image)"""
    import subprocess
    exe = _mitsuba_cli()
    for i, text in enumerate(('<scene version="0.6.0"><integrator type="path"></scene>',
                              '<scene version="0.6.0"><integrator type="no_such_integrator"/></scene>')):
        p = tmp_path / ("bad%d.xml" % i)
        p.write_text(text)
        r = subprocess.run([exe, "-q", "-p", "1", "-o", str(tmp_path / ("bad%d.pfm" % i)), str(p)], capture_output=True, text=True, cwd=str(tmp_path), timeout=120)
        assert r.returncode != 0 and not os.path.exists(tmp_path / ("bad%d.pfm" % i))


# ---- QMC samplers of SURVEY 8(f) row 4: `sobol` (the reference's stream as it stands) and `stratified` (its construction, addressable) ----
def test_sobol_sampler_of_the_reference_is_reproduced_bit_for_bit(ref, olibm):
    """PHIP_SAMPLER_SOBOL: the reference's OWN `path` integrator with its OWN `sobol` sampler plugin (src/samplers/sobol.cpp: Gruenschloss'
    per-pixel enumeration of the global Sobol' sequence, dimensions consumed in call order, the quirk that no 2D request is served from
    dimension 4) against the restatement fed with the plugin's direction numbers as data (ref_ffi.sobol_tables reads them out of the loaded
    sobol.so): every sample bit-identical -- diffuse, two-sided microfacet and dielectric scenes (non-smooth vertices make one request)."""
    gauss_libm = olibm.gaussian_filter(0.5, libm=True)
    for name, build, md, spp in (("cornell", lambda: S.cornell_box(24, 20, gauss_libm), 8, 4), ("zoo", lambda: RS.zoo(gauss_libm, None), 8, 4),
                                 ("glass", lambda: RS.glass(gauss_libm, None), 12, 2), ("envmap", lambda: RS.envmap(gauss_libm, live_mip(ref)), 6, 4),
                                 ("textures", lambda: RS.textures(gauss_libm, live_mip(ref)), 6, 4), ("const_env", lambda: RS.const_env(gauss_libm, live_mip(ref)), 6, 4)):
        desc = build().desc()
        p = A.default_render_params(spp=spp, max_depth=md, block_size=256, sobol=ref.sobol_tables(desc.film.crop_width, desc.film.crop_height))
        rs = ref.RefScene(desc)
        _, smp = rs.render(p, sampler="sobol")
        osc = olibm.OracleScene(desc, libm=True)
        _, osmp, _ = osc.render(p, want_samples=True)
        # the plugin's `scramble` (a frame number, run through TEA: sobol.cpp:92-102): the driver passes `seed` as that property
        p.seed = 7; p.sobol_scramble = ref.sobol_scramble(7)
        _, smp7 = rs.render(p, sampler="sobol")
        _, osmp7, _ = osc.render(p, want_samples=True)
        assert (smp7.view(np.uint32) == osmp7.view(np.uint32)).all() and not np.array_equal(smp7, smp), name
        rs.close(); osc.close()
        assert smp[..., :3].mean() > 0.01
        # the committed fixture (the GPU box has no reference plugin) holds the same numbers
        from conftest import sobol_tables
        for a, b in zip(sobol_tables(desc.film.crop_width, desc.film.crop_height), ref.sobol_tables(desc.film.crop_width, desc.film.crop_height, 128)):
            assert np.array_equal(a, b)
        assert (smp.view(np.uint32) == osmp.view(np.uint32)).all(), (name, float((smp.view(np.uint32) == osmp.view(np.uint32)).all(axis=-1).mean()))


def test_qmc_samplers_on_a_crop_window_off_the_films_origin(ref, olibm):
    """Round 5 (VERDICT r4 item 4 of "what's missing"): a crop window that does not start at the film's origin.  The reference's image blocks -- and with them the
    pixel positions its samplers' generate() sees -- are relative to the crop window (renderproc.cpp:163-164 starts them at (0, 0)), the resolution the sequences
    are partitioned over is the crop size (integrator.cpp:40-41): the crop OFFSET never reaches sobol.cpp:170-196 / halton.cpp:274-328.  So there is nothing to
    restate -- only a refusal to drop; the reference's own `path` + `sobol` / `halton` / `hammersley` on such a window, bit for bit."""
    gauss_libm = olibm.gaussian_filter(0.5, libm=True)
    sb = S.cornell_box(96, 80, gauss_libm)
    sb.hdrfilm(96, 80, gauss_libm, crop=(37, 22, 24, 20))
    desc = sb.desc()
    rs = ref.RefScene(desc); osc = olibm.OracleScene(desc, libm=True)
    for sampler, kw in (("sobol", dict(sobol=ref.sobol_tables(24, 20))),
                        ("halton", dict(sampler=A.PHIP_SAMPLER_HALTON, qmc=ref.qmc_tables(-1, 256), seed=0)),
                        ("hammersley", dict(sampler=A.PHIP_SAMPLER_HAMMERSLEY, qmc=ref.qmc_tables(-1, 256), seed=0))):
        p = A.default_render_params(spp=4, max_depth=6, block_size=256, **kw)
        _, smp = rs.render(p, sampler=sampler)
        _, osmp, _ = osc.render(p, want_samples=True)
        assert smp[..., :3].mean() > 0.01
        assert (smp.view(np.uint32) == osmp.view(np.uint32)).all(), (sampler, float((smp.view(np.uint32) == osmp.view(np.uint32)).all(axis=-1).mean()))
    rs.close(); osc.close()


def test_direct_with_the_reference_qmc_samplers_is_reproduced_bit_for_bit(ref, olibm):
    """SURVEY 8(f) row 4, the rest of it: the reference's OWN `direct` integrator with its OWN `sobol` / `halton` / `hammersley` plugins against the
    restatement.  `direct` requests sample ARRAYS for more than one shading sample of a kind (direct.cpp:139-146): the samplers fill them in
    generate() from dimensions 5.. of the points "sample j of this pixel" (sobol.cpp:170-196, halton.cpp:274-328), single samples are the next 2D
    requests -- (2, 3), then (5, 6): dimension 4 is never handed out.  hammersley has no arrays (hammersley.cpp:293-300): single samples only."""
    from mitsuba_amd._ffi import PhipError  # noqa: F401
    gauss_libm = olibm.gaussian_filter(0.5, libm=True)
    for name, build in (("cornell", lambda: S.cornell_box(24, 20, gauss_libm)), ("zoo", lambda: RS.zoo(gauss_libm, None)), ("glass", lambda: RS.glass(gauss_libm, None))):
        desc = build().desc()
        rs = ref.RefScene(desc); osc = olibm.OracleScene(desc, libm=True)
        w, h = desc.film.crop_width, desc.film.crop_height
        for e, b, spp in ((1, 1, 4), (3, 2, 4), (0, 2, 3), (2, 0, 4), (1, 5, 2), (4, 1, 4)):
            for smp_name, kw in (("sobol", dict(sobol=ref.sobol_tables(w, h))), ("halton", dict(sampler=A.PHIP_SAMPLER_HALTON, qmc=ref.qmc_tables(-1, 256))),
                                 ("halton", dict(sampler=A.PHIP_SAMPLER_HALTON, qmc=ref.qmc_tables(7, 256), seed=9)),
                                 ("hammersley", dict(sampler=A.PHIP_SAMPLER_HAMMERSLEY, qmc=ref.qmc_tables(-1, 256)))):
                if smp_name == "hammersley" and (e > 1 or b > 1):
                    continue
                p = A.default_render_params(spp=spp, block_size=256, integrator=A.PHIP_INTEGRATOR_DIRECT, emitter_samples=e, bsdf_samples=b, **kw)
                _, rsmp = rs.render(p, sampler=smp_name)
                _, osmp, _ = osc.render(p, threads=1, want_samples=True)
                assert rsmp[..., :3].mean() > 0.005, (name, smp_name, e, b)
                same = (rsmp.view(np.uint32) == osmp.view(np.uint32)).all(axis=-1)
                assert same.all(), (name, smp_name, e, b, float(same.mean()))
        # hammersley refuses sample arrays, like the plugin
        p = A.default_render_params(spp=4, block_size=256, integrator=A.PHIP_INTEGRATOR_DIRECT, emitter_samples=2, bsdf_samples=1, sampler=A.PHIP_SAMPLER_HAMMERSLEY, qmc=ref.qmc_tables(-1, 256))
        with pytest.raises(RuntimeError):
            osc.render(p, threads=1, want_samples=True)
        rs.close(); osc.close()


def test_stratified_construction_on_the_reference(ref, olibm):
    """PHIP_SAMPLER_STRATIFIED: the reference's `path` consuming the addressable stratified construction through the glue sampler
    (ref_glue/ctr_sampler.cpp, `stratified` mode: it calls nothing of the oracle) equals the restatement sample for sample; and the
    construction stratifies like the reference's `stratified` plugin: the Cornell box converges like the reference's own stratified render (mean absolute error at 16 spp within 25 %)."""
    gauss_libm = olibm.gaussian_filter(0.5, libm=True)
    desc = S.cornell_box(24, 20, gauss_libm).desc()
    for spp in (4, 16):
        p = A.default_render_params(spp=spp, max_depth=8, block_size=256); p.sampler = A.PHIP_SAMPLER_STRATIFIED
        rs = ref.RefScene(desc)
        _, smp = rs.render(p, sampler="ctr")
        osc = olibm.OracleScene(desc, libm=True)
        _, osmp, _ = osc.render(p, want_samples=True)
        assert (smp.view(np.uint32) == osmp.view(np.uint32)).all()
        rs.close(); osc.close()
    # convergence: error against a high-spp reference image, ours vs the reference's own `stratified` and `independent`
    desc = S.cornell_box(32, 32, gauss_libm).desc()
    rs = ref.RefScene(desc)
    truth, _ = rs.render_job(A.default_render_params(spp=4096, max_depth=4), threads=os.cpu_count(), sampler="independent")
    def mse(img): return float(np.abs(img - truth).mean())       # mean ABSOLUTE error: the squared error is dominated by a few pixels
    p16 = A.default_render_params(spp=16, max_depth=4)
    own, _ = rs.render_job(p16, threads=2, sampler="stratified")
    ind, _ = rs.render_job(p16, threads=2, sampler="independent")
    ps = A.default_render_params(spp=16, max_depth=4); ps.sampler = A.PHIP_SAMPLER_STRATIFIED
    ours, _ = rs.render_job(ps, threads=2, sampler="ctr")
    rs.close()
    print("mean abs error at 16 spp: stratified construction %.3e, the reference's stratified %.3e, independent %.3e" % (mse(ours), mse(own), mse(ind)))
    assert mse(ours) < 1.25 * mse(own) and mse(ours) < mse(ind)


def test_stratified_construction_with_direct_on_the_reference(ref, olibm):
    """Round 5 (the last piece of SURVEY 8(f) row 4): PHIP_SAMPLER_STRATIFIED with `direct`.  More than one shading sample of a kind is a requested 2D array
    (direct.cpp:139-146), which the reference's `stratified` fills with ONE Latin hypercube over all sampleCount x count entries (stratified.cpp:160-164 ->
    latinHypercube); a single one is the sample's next 2D request (one cell of the sampleCount grid).  The reference's OWN `direct` consuming the addressable
    construction through the glue sampler equals the restatement sample for sample, and the construction converges like the reference's own `direct` + `stratified`."""
    gauss_libm = olibm.gaussian_filter(0.5, libm=True)
    desc = S.cornell_box(24, 20, gauss_libm).desc()
    rs = ref.RefScene(desc); osc = olibm.OracleScene(desc, libm=True)
    for spp, e, b in ((4, 1, 1), (4, 3, 2), (16, 2, 1), (9, 1, 4), (4, 0, 3), (4, 2, 0)):
        p = A.default_render_params(spp=spp, block_size=256, integrator=A.PHIP_INTEGRATOR_DIRECT, emitter_samples=e, bsdf_samples=b); p.sampler = A.PHIP_SAMPLER_STRATIFIED
        _, smp = rs.render(p, sampler="ctr")
        _, osmp, _ = osc.render(p, want_samples=True)
        assert smp[..., :3].mean() > 0.01
        assert (smp.view(np.uint32) == osmp.view(np.uint32)).all(), (spp, e, b, float((smp.view(np.uint32) == osmp.view(np.uint32)).all(axis=-1).mean()))
    rs.close(); osc.close()
    # convergence: direct lighting of the Cornell box, 16 spp x (2 emitter + 2 BSDF samples): ours within 25 % of the reference's own stratified, better than independent
    desc = S.cornell_box(32, 32, gauss_libm).desc()
    rs = ref.RefScene(desc)
    kw = dict(integrator=A.PHIP_INTEGRATOR_DIRECT, emitter_samples=2, bsdf_samples=2)
    truth, _ = rs.render_job(A.default_render_params(spp=4096, **kw), threads=os.cpu_count(), sampler="independent")
    def mae(img): return float(np.abs(img - truth).mean())
    p16 = A.default_render_params(spp=16, **kw)
    own, _ = rs.render_job(p16, threads=2, sampler="stratified")
    ind, _ = rs.render_job(p16, threads=2, sampler="independent")
    ps = A.default_render_params(spp=16, **kw); ps.sampler = A.PHIP_SAMPLER_STRATIFIED
    ours, _ = rs.render_job(ps, threads=2, sampler="ctr")
    rs.close()
    print("direct, mean abs error at 16 spp: stratified construction %.3e, the reference's stratified %.3e, independent %.3e" % (mae(ours), mae(own), mae(ind)))
    assert mae(ours) < 1.25 * mae(own) and mae(ours) < mae(ind)


def test_halton_and_hammersley_samplers_of_the_reference_are_reproduced_bit_for_bit(ref, olibm):
    """PHIP_SAMPLER_HALTON / _HAMMERSLEY: the reference's OWN `path` with its OWN `halton` / `hammersley` sampler plugins (Gruenschloss' enumeration of the
    points per pixel over bases 2 and 3 / of the Hammersley set, the scrambled radical inverses of qmc.cpp:141-166, pixel positions modulo 128, the
    dimension bookkeeping they share with `sobol`) against the restatement fed with the reference's prime table and the digit permutations of its
    PermutationStorage as data: every sample bit-identical -- Faure permutations (the default), none, pseudorandom ones; a film wider than 128 pixels"""
    from conftest import qmc_tables
    gauss_libm = olibm.gaussian_filter(0.5, libm=True)
    for name, build, md, spp in (("cornell", lambda: S.cornell_box(24, 20, gauss_libm), 8, 4), ("zoo", lambda: RS.zoo(gauss_libm, None), 8, 4),
                                 ("glass", lambda: RS.glass(gauss_libm, None), 12, 2), ("envmap", lambda: RS.envmap(gauss_libm, live_mip(ref)), 6, 4),
                                 ("cornell wide", lambda: S.cornell_box(150, 9, gauss_libm), 5, 3)):
        desc = build().desc()
        rs = ref.RefScene(desc); osc = olibm.OracleScene(desc, libm=True)
        for smp_name, kind in (("halton", A.PHIP_SAMPLER_HALTON), ("hammersley", A.PHIP_SAMPLER_HAMMERSLEY)):
            for seed, scramble in ((0, -1), (2, 0), (9, 7)):          # (ref_driver passes `seed` on as the plugin's `scramble` property: 0 -> default, n -> n - 2)
                p = A.default_render_params(spp=spp, max_depth=md, block_size=256, sampler=kind, qmc=ref.qmc_tables(scramble, 256), seed=seed)
                _, smp = rs.render(p, sampler=smp_name)
                _, osmp, _ = osc.render(p, want_samples=True)
                assert smp[..., :3].mean() > 0.01
                assert (smp.view(np.uint32) == osmp.view(np.uint32)).all(), (name, smp_name, scramble, float((smp.view(np.uint32) == osmp.view(np.uint32)).all(axis=-1).mean()))
        rs.close(); osc.close()
    # the committed fixture (the GPU box has no reference) holds the same numbers
    for scramble in (-1, 7, 0):
        a, b = qmc_tables(scramble), ref.qmc_tables(scramble, 64)
        assert np.array_equal(a[0], b[0]) and (a[1] is None and b[1] is None or np.array_equal(a[1], b[1]))
