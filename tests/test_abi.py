"""The C-ABI shared library: loads without a GPU, exports every symbol include/phip.h declares,
struct layouts agree with the ctypes mirror, and compute entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from mitsuba_amd import _abi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "phip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(phip_[a-z_]+)\s*\(", src)))


def test_header_declares_the_boundary():
    fns = declared_functions()
    for f in ["phip_scene_create", "phip_scene_destroy", "phip_render", "phip_render_device", "phip_trace",
              "phip_cancel", "phip_last_error", "phip_device_count", "phip_develop", "phip_get_samples"]:
        assert f in fns


def test_library_exports_every_declared_symbol(phip):
    for f in declared_functions():
        assert hasattr(phip, f), "libphip.so does not export " + f


def test_product_library_does_not_export_the_test_hooks():
    """round 6 (hygiene): phip_debug_* -- host twins of device functions, the fmath probe, the calibration kernels -- live in libphip_debug.so (the same sources
    with -DPHIP_DEBUG_HOOKS=1), which only tests/ and tools/ load; the shipped library exports the boundary of include/phip.h and nothing of them"""
    import subprocess
    from mitsuba_amd import _ffi
    out = subprocess.run(["nm", "-D", "--defined-only", _ffi.LIB], capture_output=True, text=True).stdout
    assert "phip_render" in out and "phip_debug_" not in out, [l for l in out.splitlines() if "phip_debug_" in l][:5]
    dbg = subprocess.run(["nm", "-D", "--defined-only", _ffi.LIB_DEBUG], capture_output=True, text=True).stdout
    assert "phip_debug_host_camera_ray" in dbg and "phip_debug_fmath" in dbg


def test_struct_sizes_match_ctypes_mirror(phip):
    structs = [A.phip_material, A.phip_shape, A.phip_emitter, A.phip_camera, A.phip_film, A.phip_scene_desc,
               A.phip_render_params, A.phip_stats, A.phip_ray, A.phip_hit, A.phip_accel_info]
    for i, s in enumerate(structs):
        assert phip.phip_abi_sizeof(i) == C.sizeof(s), s.__name__


def test_library_does_not_link_the_oracle():
    """the product must not route through oracle/: no oracle symbol or soname in libphip.so"""
    from mitsuba_amd import _ffi
    data = open(_ffi.LIB, "rb").read()
    assert b"oracle_" not in data and b"liboracle" not in data


def test_no_gpu_fails_loudly(phip, have_gpu, gauss):
    if have_gpu:
        pytest.skip("a GPU is present")
    from mitsuba_amd import scene as S
    d = S.cornell_box(32, 32, gauss).desc()
    h = phip.phip_scene_create(C.byref(d), 0)
    assert not h
    assert b"HIP" in phip.phip_last_error() or b"device" in phip.phip_last_error()
    assert phip.phip_device_count() < 0 or phip.phip_device_count() == 0
