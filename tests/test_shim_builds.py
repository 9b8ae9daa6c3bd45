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
