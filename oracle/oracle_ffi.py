"""ORACLE -- TEST INFRASTRUCTURE ONLY.

ctypes loader for oracle/_build/liboracle.so (the CPU restatement of the reference path).
Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the
product package (mitsuba_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from mitsuba_amd import _abi as A  # noqa: E402  (struct layouts only -- no product code paths)

_libs = {}


def build(libm=False, quiet=True):
    target = "libm" if libm else "all"
    r = subprocess.run(["make", "-C", HERE, target], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)
    if not quiet:
        print(r.stdout)


def lib(libm=False):
    key = "libm" if libm else "pm"
    if key in _libs:
        return _libs[key]
    path = os.path.join(HERE, "_build", "liboracle_libm.so" if libm else "liboracle.so")
    if not os.path.exists(path):
        build(libm)
    L = C.CDLL(path)
    fp, u8p, u32 = C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.c_uint32
    L.oracle_last_error.restype = C.c_char_p
    L.oracle_scene_create.restype = C.c_void_p
    L.oracle_scene_create.argtypes = [C.POINTER(A.phip_scene_desc)]
    L.oracle_scene_destroy.argtypes = [C.c_void_p]
    L.oracle_render.argtypes = [C.c_void_p, C.POINTER(A.phip_render_params), C.c_int, C.c_int, fp, fp, C.POINTER(A.phip_stats)]
    L.oracle_render_masks.argtypes = [C.c_void_p, C.POINTER(A.phip_render_params), C.c_int, C.c_int, fp, fp, C.POINTER(C.c_uint32), C.POINTER(A.phip_stats)]
    L.oracle_trace.argtypes = [C.c_void_p, C.POINTER(A.phip_ray), C.c_size_t, C.POINTER(A.phip_hit), u8p, C.POINTER(A.phip_stats)]
    L.oracle_trace_bruteforce.argtypes = [C.c_void_p, C.POINTER(A.phip_ray), C.c_size_t, C.POINTER(A.phip_hit)]
    L.oracle_gaussian_filter.argtypes = [C.c_float, fp, fp]
    L.oracle_scene_set_bruteforce.argtypes = [C.c_void_p, C.c_int]; L.oracle_scene_set_bruteforce.restype = None
    L.oracle_path_sample.argtypes = [C.c_void_p, C.POINTER(A.phip_render_params), C.c_int, C.c_int, C.c_int, fp, C.c_int]
    L.oracle_camera_ray.argtypes = [C.c_void_p, C.c_float, C.c_float, C.POINTER(A.phip_ray)]
    L.oracle_sfmt_words.argtypes = [C.c_uint64, C.c_size_t, C.POINTER(C.c_uint64)]
    L.oracle_sfmt_floats.argtypes = [C.c_uint64, C.c_int, C.c_size_t, fp]
    L.oracle_ctr_block.argtypes = [u32, u32, u32, u32, fp]
    L.oracle_ld_point.argtypes = [u32, u32, u32, u32, u32, fp]
    L.oracle_env_sample_direct.argtypes = [C.c_void_p, fp, C.c_size_t, fp, fp, fp, fp]
    L.oracle_env_pdf_direct.argtypes = [C.c_void_p, fp, C.c_size_t, fp, fp]
    L.oracle_rinv_sample.argtypes = [C.POINTER(C.c_uint32), u32, C.POINTER(C.c_uint16), C.c_int, C.c_uint64, C.c_uint64, u32]; L.oracle_rinv_sample.restype = C.c_float
    L.oracle_clipped_aabb.argtypes = [fp, fp, fp]
    L.oracle_bsdf_sample.argtypes = [C.c_void_p, u32, C.c_size_t, fp, fp, fp, fp, fp, u8p]
    L.oracle_bsdf_eval_pdf.argtypes = [C.c_void_p, u32, C.c_size_t, fp, fp, fp, fp]
    L.oracle_mf_sample.argtypes = [C.c_int, C.c_float, C.c_float, C.c_int, C.c_size_t, fp, fp, fp, fp]
    L.oracle_mf_pdf.argtypes = [C.c_int, C.c_float, C.c_float, C.c_int, C.c_size_t, fp, fp, fp, fp]
    L.oracle_sample_emitter.argtypes = [C.c_void_p, fp, fp, C.c_size_t, fp, fp, fp, fp, fp, fp]
    L.oracle_fmath.argtypes = [C.c_int, C.c_size_t, fp, fp, fp]
    L.oracle_mip_eval.argtypes = [C.POINTER(A.phip_texture), C.c_size_t, fp, fp, fp, fp]
    L.oracle_kd_info_get.argtypes = [C.c_void_p, C.c_void_p]
    _libs[key] = L
    return L


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def gaussian_filter(stddev=0.5, libm=False):
    """(radius, table[32]) of the default reconstruction filter (gaussian.cpp:34-57, rfilter.cpp:38-57)."""
    L = lib(libm)
    r = C.c_float()
    t = (C.c_float * 32)()
    L.oracle_gaussian_filter(stddev, C.byref(r), t)
    return (r.value, list(t))


class KdInfo(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("n_indices", C.c_uint32), ("max_depth", C.c_uint32), ("retracted", C.c_uint32),
                ("exp_traversal_steps", C.c_double), ("exp_leaves_visited", C.c_double),
                ("exp_prims_intersected", C.c_double), ("sah_cost", C.c_double),
                ("aabb_min", C.c_float * 3), ("aabb_max", C.c_float * 3)]


class OracleScene:
    def __init__(self, desc, libm=False):
        self.L = lib(libm)
        self.desc = desc
        self.h = self.L.oracle_scene_create(C.byref(desc))
        if not self.h:
            raise RuntimeError("oracle_scene_create: " + self.L.oracle_last_error().decode())
        self.width, self.height = desc.film.crop_width, desc.film.crop_height

    def close(self):
        if self.h:
            self.L.oracle_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, params, threads=None, sampler="ctr", want_samples=False):
        threads = threads or os.cpu_count() or 1
        film = np.zeros((self.height, self.width, 5), np.float32)
        samples = np.zeros((self.height, self.width, params.spp, 4), np.float32) if want_samples else None
        st = A.phip_stats()
        rc = self.L.oracle_render(self.h, C.byref(params), threads, 0 if sampler == "ctr" else 1, _fp(film),
                                  _fp(samples) if want_samples else None, C.byref(st))
        if rc != 0:
            raise RuntimeError("oracle_render: " + self.L.oracle_last_error().decode())
        return film, samples, st

    def smooth_masks(self, params, threads=None):
        """per sample ([y][x][sample], ctr stream): bit d-1 = the BSDF at path vertex d has a smooth component"""
        threads = threads or os.cpu_count() or 1
        film = np.zeros((self.height, self.width, 5), np.float32)
        masks = np.zeros((self.height, self.width, params.spp), np.uint32)
        st = A.phip_stats()
        rc = self.L.oracle_render_masks(self.h, C.byref(params), threads, 0, _fp(film), None, masks.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(st))
        if rc != 0:
            raise RuntimeError("oracle_render_masks: " + self.L.oracle_last_error().decode())
        return masks

    def trace(self, rays, closest=True, shadow=False, bruteforce=False):
        n = len(rays)
        r = np.ascontiguousarray(rays, dtype=np.float32).reshape(n, 8)
        hits = np.zeros((n, 4), np.float32) if closest else None
        occ = np.zeros(n, np.uint8) if shadow else None
        st = A.phip_stats()
        rp = r.ctypes.data_as(C.POINTER(A.phip_ray))
        if bruteforce:
            self.L.oracle_trace_bruteforce(self.h, rp, n, hits.ctypes.data_as(C.POINTER(A.phip_hit)))
        else:
            self.L.oracle_trace(self.h, rp, n, hits.ctypes.data_as(C.POINTER(A.phip_hit)) if closest else None,
                                occ.ctypes.data_as(C.POINTER(C.c_uint8)) if shadow else None, C.byref(st))
        return hits, occ, st

    def kd_info(self):
        k = KdInfo()
        self.L.oracle_kd_info_get(self.h, C.byref(k))
        return k

    def set_bruteforce(self, on=True):
        """ray queries by a sweep over every triangle (the structure-independent answer) instead of the reference's kd-tree"""
        self.L.oracle_scene_set_bruteforce(self.h, 1 if on else 0)

    def path_sample(self, params, px, py, k, verbose=False):
        """(R, G, B, alpha) of ONE sample of the path tracer with the stream keys of the full frame"""
        out = np.zeros(4, np.float32)
        if self.L.oracle_path_sample(self.h, C.byref(params), px, py, k, _fp(out), 1 if verbose else 0) != 0:
            raise RuntimeError("oracle_path_sample: " + self.L.oracle_last_error().decode())
        return out

    def camera_ray(self, sx, sy):
        r = A.phip_ray()
        self.L.oracle_camera_ray(self.h, sx, sy, C.byref(r))
        return np.array(list(r.o) + [r.mint] + list(r.d) + [r.maxt], np.float32)


def develop(film):
    """RGB = sum / weight, 0 where weight == 0 (fmtconv.cpp:979-991)."""
    w = film[..., 4:5]
    with np.errstate(divide="ignore", invalid="ignore"):
        rgb = np.where(w != 0, film[..., :3] / w, 0.0)
    return rgb.astype(np.float32)


def hits_prim(hits):
    return hits.view(np.uint32)[:, 3]
