"""The drop-in, end to end: the reference's own Scene / PluginManager / RenderJob (oracle/_ref: Mitsuba 0.6
compiled from /root/reference) load the product's plugin shims `path_hip.so` / `direct_hip.so`
(mitsuba_amd/plugin/*.cpp compiled against the reference's headers, oracle/Makefile.ref `shims`) as the scene's integrator;
the shim flattens the reference's object graph through public getters, renders through the C ABI on the GPU and hands the
film back through Film::setBitmap.  Compared with (a) the same scene description rendered through the ctypes harness --
the flattening must round-trip -- and (b) the reference's own `path` / `direct` on the CPU, statistically."""
import os

import numpy as np
import pytest

from conftest import rel_l2
from mitsuba_amd import _abi as A, scene as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref():
    from oracle import ref_ffi
    if not ref_ffi.available() or not ref_ffi.have_shims():
        pytest.skip("oracle/_ref (reference build + plugin shims) is not present")
    ref_ffi.lib()
    return ref_ffi


def stock_scene(gauss, res=(96, 96)):
    """what a stock Mitsuba exposes through public getters: diffuse / dielectric / roughconductor, area + constant emitters"""
    sb = S.SceneBuilder()
    sb.constant((0.05, 0.06, 0.08))
    S.cornell_box(res[0], res[1], gauss, sb=sb)
    P, T, N = S.sphere_mesh((185, 120, 170), 70.0, 24, 12); sb.mesh(P, T, sb.dielectric(1.5, 1.0), normals=N)
    P, T, N = S.sphere_mesh((370, 330, 350), 60.0, 24, 12); sb.mesh(P, T, sb.roughconductor(alpha=0.15, eta=S.CU_ETA, k=S.CU_K), normals=N)
    return sb


def live_mip(ref):
    def mip(key, image, kind, wrap_u="repeat", wrap_v=None, filter_type="ewa", max_anisotropy=None):
        return ref.RefMip(image, kind=kind, wrap_u=wrap_u, wrap_v=wrap_v, filter_type=filter_type, max_anisotropy=max_anisotropy).levels
    return mip


def scenes(ref, gauss):
    import ref_scenes as RS
    yield "stock", stock_scene(gauss).desc()
    # two-sided BSDFs, bitmap textures (the plugin's own Lanczos / half-precision pyramid), the envmap: the shim reads them
    # out of the reference's plugin objects (PHIP_REFERENCE_SOURCES)
    for build in (RS.zoo, RS.textures, RS.roughness_maps, RS.envmap, RS.const_env, RS.atrium):
        yield build.__name__, build(gauss, live_mip(ref)).desc()


def test_path_hip_plugin_inside_the_reference(phip, ref, oracle, gauss):
    if phip.phip_device_count() <= 0:
        pytest.fail("no HIP device visible")
    for name, desc in scenes(ref, gauss):
        check_scene(ref, name, desc)


def check_scene(ref, name, desc):
    from mitsuba_amd.integrator import Scene, PathHIP, DirectHIP, VolPathSimpleHIP, HDRFilm
    rs = ref.RefScene(desc)
    gs = Scene(desc)
    for plugin, Integ, kw, rkw, sampler in (("path_hip", PathHIP, dict(maxDepth=6), dict(max_depth=6), "independent"),
                                           ("volpath_simple_hip", VolPathSimpleHIP, dict(maxDepth=6), dict(max_depth=6, integrator=A.PHIP_INTEGRATOR_VOLPATH_SIMPLE), "independent"),
                                           ("path_hip", PathHIP, dict(maxDepth=6), dict(max_depth=6), "ldsampler"),
                                           ("direct_hip", DirectHIP, dict(emitterSamples=2, bsdfSamples=2),
                                            dict(integrator=A.PHIP_INTEGRATOR_DIRECT, emitter_samples=2, bsdf_samples=2), "independent"),
                                           ("direct_hip", DirectHIP, dict(emitterSamples=2, bsdfSamples=1),
                                            dict(integrator=A.PHIP_INTEGRATOR_DIRECT, emitter_samples=2, bsdf_samples=1), "ldsampler")):
        # <sampler type="ldsampler"/> in the scene makes the shim select PHIP_SAMPLER_LD (phip_flatten.h: checkSampler)
        ld = dict(sampler=A.PHIP_SAMPLER_LD) if sampler == "ldsampler" else {}
        p = A.default_render_params(spp=32, **rkw)
        img, sec = rs.render_job(p, threads=2, plugin=plugin, sampler=sampler)   # Mitsuba -> plugin shim -> libphip.so -> GPU
        film = HDRFilm(gs.width, gs.height)
        assert Integ(**kw).render(gs, film, 32, **ld)                       # ctypes harness -> libphip.so -> GPU
        direct = film.develop()
        assert np.isfinite(img).all() and img.max() > 0
        r = rel_l2(img, direct)
        print("%s: %s inside Mitsuba (%s) vs ctypes harness: rel L2 %.3e (%.3f s)" % (name, plugin, sampler, r, sec))
        assert r < 1e-5                                                      # same scene after the round trip through Mitsuba's objects
        cpu, _ = rs.render_job(p, threads=8, sampler=sampler)                # the reference's own integrator (and sampler) on the CPU
        dm = abs(img.mean() - cpu.mean()) / cpu.mean()
        print("    vs the reference's CPU %s: mean differs by %.2f %%, rel L2 %.2f" % (plugin.replace("_hip", ""), 100 * dm, rel_l2(img, cpu)))
        noisy = plugin == "volpath_simple_hip"                               # (no multiple importance sampling: fireflies on the atrium's copper at 32 spp)
        if noisy:
            # the mean of such an image is a few fireflies (the reference's own two runs differ by 5-15 %: its `independent` streams depend on which thread
            # takes which block): the means are compared with both images clipped at the reference's 99th percentile
            q = float(np.percentile(cpu, 99.0))
            dq = abs(np.minimum(img, q).mean() - np.minimum(cpu, q).mean()) / np.minimum(cpu, q).mean()
            print("    clipped at the reference's 99th percentile: mean differs by %.2f %%" % (100 * dq))
            ok = dm < 0.3 and dq < 0.15 and rel_l2(img, cpu) < 1.5       # (measured over repeated runs, profiles/r05_gpu_call_v_*: atrium 5.1 / 7.2 % clipped, 7.0 / 13.2 % raw; the others < 3 %)
        else:
            ok = dm < 0.08 and rel_l2(img, cpu) < 0.6                        # small, noisy images (32 spp): two independent renders
        if not ok:
            rs.close(); gs.close()                                           # (the reference's worker threads must not outlive a failure: later tests time frames)
        assert ok, (name, plugin, sampler, float(dm), float(rel_l2(img, cpu)))
    rs.close(); gs.close()


def test_qmc_samplers_of_the_scene_reach_the_device(phip, ref, oracle, gauss):
    """<sampler type="sobol"/>, "halton", "hammersley" and "stratified" (SURVEY 8(f) row 4).  The first three are deterministic, so the comparison is the strongest
    of this file: Mitsuba's own `path` + `sobol` on the CPU and path_hip + the same <sampler> on the GPU render THE SAME IMAGE (<= 1e-3 rel L2;
    the shim reads the direction numbers out of the loaded sobol plugin, phip_flatten.h: setSobol) -- no glue sampler anywhere.  `stratified`
    keeps its stratification (the device's stratified stream) and agrees with the harness."""
    import ref_scenes as RS
    from conftest import sobol_tables
    from mitsuba_amd.integrator import Scene, PathHIP, HDRFilm
    if phip.phip_device_count() <= 0:
        pytest.fail("no HIP device visible")
    for name, desc in (("stock", stock_scene(gauss).desc()), ("zoo", RS.zoo(gauss, None).desc()), ("cornell", S.cornell_box(96, 64, gauss).desc())):
        rs = ref.RefScene(desc); gs = Scene(desc)
        w, h = desc.film.crop_width, desc.film.crop_height
        p = A.default_render_params(spp=16, max_depth=6)
        from conftest import qmc_tables
        for sampler, kw in (("sobol", dict(sobol=sobol_tables(w, h))), ("stratified", dict(sampler=A.PHIP_SAMPLER_STRATIFIED)),
                            ("halton", dict(sampler=A.PHIP_SAMPLER_HALTON, qmc=qmc_tables(-1))), ("hammersley", dict(sampler=A.PHIP_SAMPLER_HAMMERSLEY, qmc=qmc_tables(-1)))):
            img, sec = rs.render_job(p, threads=2, plugin="path_hip", sampler=sampler)     # Mitsuba -> plugin shim -> libphip.so -> GPU
            film = HDRFilm(gs.width, gs.height)
            assert PathHIP(maxDepth=6).render(gs, film, 16, **kw)                           # ctypes harness -> libphip.so -> GPU
            r = rel_l2(img, film.develop())
            cpu, _ = rs.render_job(p, threads=4, sampler=sampler)                           # the reference's own path + its own sampler plugin
            rc = rel_l2(img, cpu)
            print("%s, %s: path_hip inside Mitsuba vs harness rel L2 %.3e; vs the reference's own path + %s on the CPU rel L2 %.3e" % (name, sampler, r, sampler, rc))
            assert r < 1e-5
            assert rc <= (0.6 if sampler == "stratified" else 1e-3), rc
        # `direct` on the deterministic samplers: sample arrays (shadingSamples 3) and single samples -- direct_hip inside Mitsuba = the reference's direct + sampler on the CPU
        from mitsuba_amd.integrator import DirectHIP
        for sampler, kw, (e, b) in (("sobol", dict(sobol=sobol_tables(w, h)), (3, 3)), ("halton", dict(sampler=A.PHIP_SAMPLER_HALTON, qmc=qmc_tables(-1)), (2, 1)),
                                    ("hammersley", dict(sampler=A.PHIP_SAMPLER_HAMMERSLEY, qmc=qmc_tables(-1)), (1, 1))):
            pd = A.default_render_params(spp=8, integrator=A.PHIP_INTEGRATOR_DIRECT, emitter_samples=e, bsdf_samples=b)
            img, sec = rs.render_job(pd, threads=2, plugin="direct_hip", sampler=sampler)
            film = HDRFilm(gs.width, gs.height)
            assert DirectHIP(emitterSamples=e, bsdfSamples=b).render(gs, film, 8, **kw)
            r = rel_l2(img, film.develop())
            cpu, _ = rs.render_job(pd, threads=4, sampler=sampler)
            rc = rel_l2(img, cpu)
            print("%s, %s: direct_hip (%d, %d) inside Mitsuba vs harness rel L2 %.3e; vs the reference's own direct + %s on the CPU rel L2 %.3e" % (name, sampler, e, b, r, sampler, rc))
            assert r < 1e-5 and rc <= 1e-3, (r, rc)
        rs.close(); gs.close()


def test_gpu_against_the_reference_on_the_same_samples(phip, ref, oracle, gauss):
    """BASELINE.json north_star, literally: "output radiance matches the reference CPU `path` integrator on the same
    scene / seed ... <= 1e-3 relative L2 at equal spp".  The reference's own `path` (and `direct`) run on the host with the
    parity stream (oracle/ref_glue/ctr_sampler.cpp: the reference consumes the random numbers the GPU consumes -- the stream is defined
    by call order, so nothing of the oracle is involved, dielectrics or not); the GPU
    renders the same scene through the C ABI.  Compared sample by sample: the two differ only in the transcendentals (libm there, phip_fmath.h
    here, <= 4 ulp), so most samples agree to the last bits, the rest to ~1e-6 -- except the handful of paths in which such
    an ulp flips a discrete decision (a Russian-roulette test, a CDF bin, a tie) and the path goes elsewhere: those are
    counted.  Path tracing is chaotic in exactly this sense; the reference compiled with another libm would differ from
    itself the same way."""
    import ref_scenes as RS
    from mitsuba_amd.integrator import Scene, PathHIP, DirectHIP, HDRFilm
    for name, desc, spp, bar in (("cornell 128x128", S.cornell_box(128, 128, gauss).desc(), 64, 1e-3),
                                 ("textures 48x32", RS.textures(gauss, live_mip(ref)).desc(), 64, 1e-3),
                                 ("roughness maps 48x32", RS.roughness_maps(gauss, live_mip(ref)).desc(), 64, 1e-3),
                                 ("atrium 160x90", S.atrium(160, 90, gauss, detail=0.5).desc(), 32, 1e-3),
                                 ("glass room 160x90", S.glass_room(160, 90, gauss, detail=0.5).desc(), 32, 1e-3),
                                 ("material zoo 32x32", RS.zoo(gauss, None).desc(), 64, 1e-3),
                                 ("envmap 40x24", RS.envmap(gauss, live_mip(ref)).desc(), 64, 1e-3)):
        rs = ref.RefScene(desc)
        gs = Scene(desc)
        for what, Integ, kw, rkw in (("path", PathHIP, dict(maxDepth=8), dict(max_depth=8)),
                                     ("direct", DirectHIP, dict(emitterSamples=2, bsdfSamples=2),
                                      dict(integrator=A.PHIP_INTEGRATOR_DIRECT, emitter_samples=2, bsdf_samples=2))):
            p = A.default_render_params(spp=spp, block_size=256, **rkw)
            rfilm, rsmp = rs.render(p, sampler="ctr")   # the reference's Li, sample by sample
            integ = Integ(**kw)
            film = HDRFilm(gs.width, gs.height)
            assert integ.render(gs, film, spp, flags=A.PHIP_FLAG_SAMPLE_BUFFER)
            gsmp = integ.samples(gs, spp)
            identical = (gsmp.view(np.uint32) == rsmp.view(np.uint32)).all(-1)
            close = (np.abs(gsmp - rsmp) <= 1e-4 * np.maximum(1.0, np.abs(rsmp))).all(-1)
            g = film.develop()
            w = rfilm[..., 4:5]
            c = np.where(w != 0, rfilm[..., :3] / np.where(w != 0, w, 1), 0)
            r = rel_l2(g, c)
            print("%s, %s, %d spp: GPU vs the reference's own %s on the same samples: %.2f %% bit-identical, %d of %d samples took another path, image rel L2 %.2e"
                  % (name, what, spp, what, 100 * identical.mean(), int((~close).sum()), close.size, r))
            assert (~close).mean() < 2e-3
            assert r <= bar
        rs.close(); gs.close()


def test_baseline_config_c1_against_the_reference(phip, ref, gauss):
    """BASELINE.json configs[0] -- "Cornell box, 256x256, 16 spp, `path` maxDepth=4, diffuse-only, CPU reference" -- rendered
    by the reference itself (its RenderJob on 8 LocalWorkers, parity-stream sampler, its HDRFilm) and by path_hip on the
    GPU: the developed images agree to <= 1e-3 relative L2 (measured ~1e-6: a few ulp-level sample differences)"""
    from mitsuba_amd.integrator import Scene, PathHIP, HDRFilm
    desc = S.cornell_box(256, 256, gauss).desc()
    rs = ref.RefScene(desc)
    cpu, sec = rs.render_job(A.default_render_params(spp=16, max_depth=4), threads=8, sampler="ctr")
    gs = Scene(desc)
    film = HDRFilm(256, 256)
    integ = PathHIP(maxDepth=4)
    assert integ.render(gs, film, 16)
    g = film.develop()
    r = rel_l2(g, cpu)
    print("C1: GPU %.1f ms vs the reference %.0f ms on 8 threads; image rel L2 %.2e, max abs diff %.2e" % (integ.stats.render_ms, 1e3 * sec, r, np.abs(g - cpu).max()))
    assert r <= 1e-3
    rs.close(); gs.close()


def test_analytic_shapes_reach_the_gpu_through_createTriMesh(phip, ref, gauss):
    """the reference's analytic `rectangle` shapes (the classic Cornell box's light and walls): the plugin shim flattens
    them through Shape::createTriMesh (rectangle.cpp:170-203: a two-triangle mesh with normals and texture coordinates) and
    keeps the area emitter attached to the right shape.  The reference's CPU `path` intersects and samples the analytic
    rectangles; the shading frames (UV tangents instead of coordinateSystem) and the light's (u,v) -> point map differ from
    the mesh, so the comparison is statistical."""
    from mitsuba_amd.integrator import Scene, PathHIP, HDRFilm
    desc = S.cornell_box(96, 96, gauss).desc()
    rs = ref.RefScene(desc, analytic_rectangles=True)
    p = A.default_render_params(spp=256, max_depth=6)
    gpu, _ = rs.render_job(p, threads=8, plugin="path_hip")
    cpu, _ = rs.render_job(p, threads=8)
    gs = Scene(desc); film = HDRFilm(96, 96); assert PathHIP(maxDepth=6).render(gs, film, 256)
    mesh = film.develop()
    for what, other in (("the reference's CPU path on the analytic shapes", cpu), ("path_hip on the triangle-mesh description", mesh)):
        dm = abs(gpu.mean() - other.mean()) / other.mean()
        r = rel_l2(gpu, other)
        print("path_hip inside Mitsuba (analytic rectangles) vs %s: mean differs by %.2f %%, rel L2 %.3f" % (what, 100 * dm, r))
        assert dm < 0.02 and r < 0.1                 # independent 256-spp renders: noise
    rs.close(); gs.close()


@pytest.mark.skipif(not os.environ.get("PHIP_FUZZ_REFERENCE"), reason="opt-in (PHIP_FUZZ_REFERENCE=1): on a many-core host the reference's "
                    "kd-tree builder spawns one thread per core for every scene, ~1.5 s per scene on the 256-thread GPU box")
def test_random_scenes_gpu_against_the_reference_on_the_same_samples(phip, ref, oracle, gauss):
    """(opt-in; written at the very end of round 1 and not yet run to completion on a GPU box -- the same comparison on fixed
    scenes is test_gpu_against_the_reference_on_the_same_samples above)
    the fuzz scenes (ref_scenes.random_scene, with the reference's own MIP pyramids): GPU against Mitsuba's `path` /
    `direct` fed with the parity stream, sample by sample.  Only glibc's rounding separates the two (HISTORY.md 3.6):
    nearly all samples agree to the last bit, the rest to ~1e-6, a handful of paths per scene at most take another branch"""
    import ref_scenes as RS
    from mitsuba_amd.integrator import Scene, PathHIP, DirectHIP, HDRFilm
    n_scenes = int(os.environ.get("PHIP_FUZZ_SCENES", "40"))
    tot = ident = diverged = 0
    for seed in range(n_scenes):
        sb, kw = RS.random_scene(gauss, seed, res=(40, 28), mip=live_mip(ref))
        desc = sb.desc()
        if kw.get("integrator") == A.PHIP_INTEGRATOR_DIRECT:
            integ = DirectHIP(emitterSamples=kw["emitter_samples"], bsdfSamples=kw["bsdf_samples"], strictNormals=bool(kw["strict_normals"]))
        else:
            integ = PathHIP(maxDepth=kw["max_depth"], rrDepth=kw["rr_depth"], strictNormals=bool(kw["strict_normals"]), hideEmitters=bool(kw["hide_emitters"]))
        gs = Scene(desc); gs.setBlockSize(64)
        film = HDRFilm(gs.width, gs.height)
        spp = 4
        assert integ.render(gs, film, spp, flags=A.PHIP_FLAG_SAMPLE_BUFFER)
        gsmp = integ.samples(gs, spp)
        p = integ.params(gs, spp)
        rs = ref.RefScene(desc)
        _, rsmp = rs.render(p, sampler="ctr")                   # (no oracle in the loop: the parity stream is defined by call order)
        both_nan = np.isnan(gsmp) & np.isnan(rsmp)
        same = ((gsmp.view(np.uint32) == rsmp.view(np.uint32)) | both_nan).all(-1)
        close = ((np.abs(gsmp - rsmp) <= 1e-4 * np.maximum(1.0, np.abs(rsmp))) | both_nan).all(-1)
        tot += same.size; ident += int(same.sum()); diverged += int((~close).sum())
        assert (~close).mean() < 5e-3, (seed, kw, float((~close).mean()))
        rs.close(); gs.close()
    print("GPU vs Mitsuba on the same samples over %d random scenes: %.2f %% of %d samples bit-identical, %d took another path"
          % (n_scenes, 100.0 * ident / tot, tot, diverged))
    assert ident / tot > 0.9


def _fullsize(tmp_path, keys, preload):
    """tools/fullsize_vs_reference.py in a subprocess (LD_PRELOAD has to be in place before the process starts)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    # the arbiter's libm is the one of the container that built oracle/_ref (oracle/ref_ffi.py: pin_libm), on every lease: VERDICT r4, item 7
    pinned = os.path.join(root, "oracle", "_ref", "pinned_libm")
    if os.path.exists(os.path.join(pinned, "libm.so.6")):
        env["LD_LIBRARY_PATH"] = pinned + (":" + env["LD_LIBRARY_PATH"] if env.get("LD_LIBRARY_PATH") else "")
    if preload:
        r = subprocess.run(["make", "-C", os.path.join(root, "oracle"), "crm"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        env["LD_PRELOAD"] = os.path.join(root, "oracle", "_build", "libcrm.so")
    out = tmp_path / ("fullsize_%s.json" % ("cr" if preload else "glibc"))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fullsize_vs_reference.py"), str(out)] + list(keys),
                       capture_output=True, text=True, env=env, timeout=3000)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return json.load(open(out))


def test_baseline_configs_at_full_size_against_the_reference(phip, ref, gauss, tmp_path):
    """BASELINE.json configs[1] and [2] AT FULL SIZE -- Cornell box 1024x1024x256 spp and the Sponza-class atrium 1920x1080x64 spp --
    rendered by path_hip on the GPU and by Mitsuba 0.6 itself (its RenderJob on every host core, parity-stream sampler): the
    developed images agree within the north star's 1e-3 relative L2.  Then the same against the reference with the correctly
    rounded transcendentals of include/phip_fmath.h LD_PRELOADed over glibc's (oracle/ref_glue/crlibm_shim.cpp), which removes
    the libm term: what is left is the handful of samples on which the reference's kd-tree returns another closest hit than a
    sweep over all triangles (tests/test_gpu_parity.py::test_c2_at_full_size_against_the_oracle, profiles/r02_c2_fullsize_sample_parity.json:
    20 of C2's 268 M samples) and the order of the float additions in the film.
    configs[3] -- the glass room, 1920x1080, maxDepth 16 -- runs at its full frame and depth with 64 of its 512 samples per pixel (the
    reference needs about a minute for that on 256 threads, eight for the full count: PHIP_FULLSIZE_C4=1 renders all 512; round 2's
    one-off run of it: 8.2e-5)."""
    keys = ["C2", "C3", "C4res"] + (["C4full"] if os.environ.get("PHIP_FULLSIZE_C4") else [])
    stock = _fullsize(tmp_path, keys + ["C2sobol"], preload=False)
    # C2 with <sampler type="sobol"/>: the reference's own sampler plugin on the CPU side, its direction numbers on the GPU side -- no parity
    # sampler anywhere, the image the reference renders for that scene file
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pinned = os.path.join(root, "oracle", "_ref", "pinned_libm")
    if os.path.exists(os.path.join(pinned, "libm.so.6")):
        built_with = open(os.path.join(pinned, "glibc_version.txt")).read().strip()
        for name, r in stock.items():
            assert os.path.realpath(r["libm_mapped"]) == os.path.realpath(os.path.join(pinned, "libm.so.6")), r["libm_mapped"]
        print("the reference's libm: %s (glibc %s of the container that built oracle/_ref; this box runs glibc %s)"
              % (os.path.relpath(pinned, root), built_with, next(iter(stock.values()))["glibc_version"]))
    for name, r in stock.items():
        print("%s vs Mitsuba 0.6 (glibc): rel L2 %.3e, %.4f %% of the pixels differ by more than 1e-3; GPU %.3f s, reference %.1f s on %d threads"
              % (name, r["rel_l2"], 100 * r["pixels_differing_by_more_than_1e-3"], r["gpu_seconds"], r["reference_seconds"], r["reference_threads"]))
        assert r["rel_l2"] <= 1e-3, (name, r)
    cr = _fullsize(tmp_path, keys, preload=True)
    for name, r in cr.items():
        print("%s vs Mitsuba 0.6 with phip_fmath.h transcendentals: rel L2 %.3e, %.4f %% of the pixels differ by more than 1e-3"
              % (name, r["rel_l2"], 100 * r["pixels_differing_by_more_than_1e-3"]))
        assert "phip_fmath" in r["reference_libm"]
        assert r["rel_l2"] <= 1e-3 and r["rel_l2"] <= 1.05 * stock[name]["rel_l2"] + 1e-6, (name, r, stock[name])
    record = os.environ.get("PHIP_FULLSIZE_RECORD")
    if record:
        import json
        json.dump({"glibc": stock, "phip_fmath_preloaded": cr}, open(record, "w"), indent=1)


def test_path_hip_in_a_scene_file_through_the_reference_cli(phip, ref, gauss, tmp_path):
    """north_star: "the new 'path_hip' integrator drops into the existing mitsuba CLI".  A scene FILE names <integrator type="path_hip"/>
    (and type="direct_hip"); the reference's own command-line front end (oracle/_ref/mitsuba = src/mitsuba/mitsuba.cpp) loads it with the
    reference's own SceneHandler, its PluginManager opens path_hip.so, its RenderJob calls the shim's render(), the GPU renders, its
    HDRFilm develops and writes the image: the same frame as the ctypes harness renders from the same description (the film passes
    through Film::setBitmap and the PFM writer: rel. L2 < 1e-5), and the frame of the reference's CPU integrator up to Monte-Carlo noise."""
    import subprocess
    import xml_scene as X
    from mitsuba_amd.integrator import Scene, PathHIP, DirectHIP, HDRFilm
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "oracle", "_ref", "mitsuba")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/mitsuba is not built")
    if phip.phip_device_count() <= 0:
        pytest.fail("no HIP device visible")
    spp = 32
    for name, desc in (("cornell", S.cornell_box(96, 80, gauss).desc()), ("stock", stock_scene(gauss).desc())):
        for plugin, Integ, kw, props in (("path_hip", PathHIP, dict(maxDepth=6), dict(maxDepth=6, rrDepth=5)),
                                         ("direct_hip", DirectHIP, dict(emitterSamples=2, bsdfSamples=2), dict(emitterSamples=2, bsdfSamples=2))):
            out = tmp_path / (name + "_" + plugin)
            xml = X.write_scene_xml(desc, str(out), integrator=plugin, integrator_props=props, sampler="independent", spp=spp)
            r = subprocess.run([exe, "-q", "-p", "2", "-o", str(out / "gpu.pfm"), xml], capture_output=True, text=True, cwd=str(out), timeout=900)
            assert r.returncode == 0 and os.path.exists(out / "gpu.pfm"), r.stdout[-3000:] + r.stderr[-3000:]
            cli = X.read_pfm(str(out / "gpu.pfm"))
            gs = Scene(desc); film = HDRFilm(gs.width, gs.height)
            assert Integ(**kw).render(gs, film, spp)
            direct = film.develop(); gs.close()
            rr = rel_l2(cli, direct)
            print("%s: <integrator type=\"%s\"/> through the reference's CLI vs the ctypes harness: rel L2 %.3e" % (name, plugin, rr))
            # (the stock scene's spheres carry vertex normals, which the OBJ loader renormalises: last-bit differences in the shading frames)
            assert np.isfinite(cli).all() and cli.max() > 0 and rr < (1e-5 if name == "cornell" else 1e-3)
            # the same file with the reference's own integrator on the CPU
            cpu_xml = X.write_scene_xml(desc, str(out / "cpu"), integrator=plugin.replace("_hip", ""), integrator_props=props, sampler="independent", spp=spp)
            r = subprocess.run([exe, "-q", "-p", "8", "-o", str(out / "cpu" / "cpu.pfm"), cpu_xml], capture_output=True, text=True, cwd=str(out / "cpu"), timeout=900)
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
            cpu = X.read_pfm(str(out / "cpu" / "cpu.pfm"))
            dm = abs(cli.mean() - cpu.mean()) / cpu.mean()
            print("    vs <integrator type=\"%s\"/> on the CPU: mean differs by %.2f %%, rel L2 %.2f" % (plugin.replace("_hip", ""), 100 * dm, rel_l2(cli, cpu)))
            assert dm < 0.08 and rel_l2(cli, cpu) < 0.6
    # the scene file as a user of the reference would write it for a QMC render: meshes in Mitsuba's own .serialized format (written by the
    # reference's TriMesh::serialize, read by its `serialized` plugin), <sampler type="sobol"/> -- path_hip and the reference's path render
    # the same image from the same file, nothing of the test harness in between
    desc = stock_scene(gauss).desc()
    imgs = {}
    for plugin in ("path_hip", "path"):
        out = tmp_path / ("sobol_" + plugin)
        xml = X.write_scene_xml(desc, str(out), integrator=plugin, integrator_props=dict(maxDepth=6, rrDepth=5), sampler="sobol", spp=16, mesh_format="serialized")
        r = subprocess.run([exe, "-q", "-p", "4", "-o", str(out / "img.pfm"), xml], capture_output=True, text=True, cwd=str(out), timeout=900)
        assert r.returncode == 0 and os.path.exists(out / "img.pfm"), r.stdout[-3000:] + r.stderr[-3000:]
        imgs[plugin] = X.read_pfm(str(out / "img.pfm"))
    rr = rel_l2(imgs["path_hip"], imgs["path"])
    print("scene.xml with .serialized meshes and <sampler type=\"sobol\"/>: mitsuba with path_hip vs mitsuba with path: rel L2 %.3e" % rr)
    assert rr <= 1e-3
