"""The drop-in, end to end: the reference's own Scene / PluginManager / RenderJob (oracle/_ref: Mitsuba 0.6
compiled from /root/reference) load the product's plugin shims `path_hip.so` / `direct_hip.so`
(mitsuba_amd/plugin/*.cpp compiled against the reference's headers, oracle/Makefile.ref `shims`) as the scene's integrator;
the shim flattens the reference's object graph through public getters, renders through the C ABI on the GPU and hands the
film back through Film::setBitmap.  Compared with (a) the same scene description rendered through the ctypes harness --
the flattening must round-trip -- and (b) the reference's own `path` / `direct` on the CPU, statistically."""
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
    for build in (RS.zoo, RS.textures, RS.envmap, RS.const_env, RS.atrium):
        yield build.__name__, build(gauss, live_mip(ref)).desc()


def test_path_hip_plugin_inside_the_reference(phip, ref, oracle, gauss):
    if phip.phip_device_count() <= 0:
        pytest.fail("no HIP device visible")
    for name, desc in scenes(ref, gauss):
        check_scene(ref, name, desc)


def check_scene(ref, name, desc):
    from mitsuba_amd.integrator import Scene, PathHIP, DirectHIP, HDRFilm
    rs = ref.RefScene(desc)
    gs = Scene(desc)
    for plugin, Integ, kw, rkw in (("path_hip", PathHIP, dict(maxDepth=6), dict(max_depth=6)),
                                   ("direct_hip", DirectHIP, dict(emitterSamples=2, bsdfSamples=2),
                                    dict(integrator=A.PHIP_INTEGRATOR_DIRECT, emitter_samples=2, bsdf_samples=2))):
        p = A.default_render_params(spp=32, **rkw)
        img, sec = rs.render_job(p, threads=2, plugin=plugin)               # Mitsuba -> plugin shim -> libphip.so -> GPU
        film = HDRFilm(gs.width, gs.height)
        assert Integ(**kw).render(gs, film, 32)                             # ctypes harness -> libphip.so -> GPU
        direct = film.develop()
        assert np.isfinite(img).all() and img.max() > 0
        r = rel_l2(img, direct)
        print("%s: %s inside Mitsuba vs ctypes harness: rel L2 %.3e (%.3f s)" % (name, plugin, r, sec))
        assert r < 1e-5                                                      # same scene after the round trip through Mitsuba's objects
        cpu, _ = rs.render_job(p, threads=8)                                 # the reference's own integrator on the CPU
        dm = abs(img.mean() - cpu.mean()) / cpu.mean()
        print("    vs the reference's CPU %s: mean differs by %.2f %%, rel L2 %.2f" % (plugin.replace("_hip", ""), 100 * dm, rel_l2(img, cpu)))
        assert dm < 0.08                                                     # small, noisy images (32 spp)
        assert rel_l2(img, cpu) < 0.6                                        # two independent 32-spp renders
    rs.close(); gs.close()
