"""The Mitsuba plugin shims (mitsuba_amd/plugin/*.cpp) must keep compiling against the reference's headers in all three
build modes of phip_flatten.h -- stock, -DPHIP_REFERENCE_ACCESSORS (the maintainer's accessor patch) and
-DPHIP_REFERENCE_SOURCES (what oracle/Makefile.ref builds and tests/test_gpu_dropin.py runs on the GPU).  Needs the
reference tree; syntax-only, in parallel."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MTS_REFERENCE", "/root/reference")
FLAGS = ("-include unistd.h -include cassert -include cstring -std=c++17 -fPIC -w -DSINGLE_PRECISION -DSPECTRUM_SAMPLES=3 -DMTS_SSE "
         "-fopenmp -I%s/include -I%s/src/integrators/path -I%s/oracle/ref_shims -I%s/include -fsyntax-only" % (REF, REF, ROOT, ROOT)).split()
MODES = {"stock": [], "accessors": ["-DPHIP_REFERENCE_ACCESSORS"], "sources": ["-DPHIP_REFERENCE_SOURCES", "-fno-access-control"]}


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "include", "mitsuba")), reason="the reference tree is not present")
def test_plugin_shims_compile_against_the_reference_in_every_mode():
    jobs = [(m, f) for m in MODES for f in ("path_hip.cpp", "direct_hip.cpp")]

    def compile_one(job):
        mode, f = job
        r = subprocess.run(["g++"] + MODES[mode] + FLAGS + [os.path.join(ROOT, "mitsuba_amd", "plugin", f)], capture_output=True, text=True)
        return mode, f, r.returncode, r.stderr[-2000:]
    with ThreadPoolExecutor(len(jobs)) as ex:
        results = list(ex.map(compile_one, jobs))
    bad = [r for r in results if r[2] != 0]
    assert not bad, "\n".join("%s %s:\n%s" % (m, f, err) for m, f, _, err in bad)


def test_stock_mode_plugin_refuses_what_it_cannot_see_into(gauss):
    """the mode of the shim a maintainer of an UNPATCHED Mitsuba builds (public headers only, no plugin sources, no -fno-access-control:
    oracle/_ref/plugins/path_hip_stock.so) must say so when a scene needs a plugin-local member it cannot reach -- the children of `twosided`, a bitmap texture --
    BEFORE anything is handed to the device, and must get past that point on a scene it can serve (where this CPU-only box then stops it: no HIP device)."""
    import numpy as np
    from mitsuba_amd import _abi as A, scene as S
    from oracle import ref_ffi as R
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "plugins", "path_hip_stock.so")):
        pytest.skip("oracle/_ref/plugins/path_hip_stock.so is not built (make -C oracle -f Makefile.ref shims)")
    R.lib()

    def error_of(sb):
        rs = R.RefScene(sb.desc())
        try:
            rs.render_job(A.default_render_params(spp=1, max_depth=3), threads=2, want_image=False, plugin="path_hip_stock")
        except Exception as e:
            return str(e)
        finally:
            rs.close()
        return ""

    two = S.cornell_box(32, 32, gauss, short_bsdf=lambda b: b.twosided(b.diffuse((0.5, 0.5, 0.5))))
    assert "twosided adapter needs -DPHIP_REFERENCE_SOURCES" in error_of(two)
    tex = S.SceneBuilder()
    img = np.random.default_rng(3).uniform(0.1, 0.9, (8, 8, 3)).astype(np.float32)
    tex.quad((0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), tex.diffuse(texture=tex.bitmap(img)), facing=(0, 0, -1), uvs=[(0, 0), (1, 0), (1, 1), (0, 1)])
    tex.quad((0, 0, -2), (1, 0, -2), (1, 1, -2), (0, 1, -2), tex.diffuse((0, 0, 0)), facing=(0, 0, 1), radiance=(5, 5, 5))
    tex.perspective((0.5, 0.5, -1.5), (0.5, 0.5, 0), (0, 1, 0), 45.0); tex.hdrfilm(32, 32, gauss)
    assert "textured reflectance needs -DPHIP_REFERENCE_SOURCES" in error_of(tex)
    # a scene the stock build serves (constant reflectances, no adapter): it passes the conversion; without a GPU the device refuses it, with one it renders
    plain = error_of(S.cornell_box(32, 32, gauss))
    assert "PHIP_REFERENCE_SOURCES" not in plain, plain
