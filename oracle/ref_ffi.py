"""ORACLE -- TEST INFRASTRUCTURE ONLY.

ctypes loader for oracle/_ref/libmtsref.so: the REFERENCE's own libcore + librender + plugins, compiled in place from
/root/reference by oracle/Makefile.ref, behind oracle/ref_driver.cpp.  Used by tests/test_ref_pin.py to pin the oracle's
restatement to the real code and by tests/golden/make_golden_ref.py to produce the committed fixtures.  It only exists
where /root/reference does (the build container); the GPU box has the prebuilt oracle/_ref at most."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from mitsuba_amd import _abi as A  # noqa: E402  (struct layouts only)

REFERENCE = os.environ.get("MTS_REFERENCE", "/root/reference")
LIB = os.path.join(HERE, "_ref", "libmtsref.so")
_lib = None


def available():
    return os.path.exists(LIB) or os.path.isdir(os.path.join(REFERENCE, "src", "libcore"))


def build(quiet=True):
    if not os.path.isdir(os.path.join(REFERENCE, "src", "libcore")):
        if os.path.exists(LIB):
            return
        raise RuntimeError("the reference tree is not present (%s) and oracle/_ref is not built" % REFERENCE)
    r = subprocess.run(["make", "-C", HERE, "-f", "Makefile.ref", "-j%d" % min(64, os.cpu_count() or 4), "REF=" + REFERENCE],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("reference build failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    if not quiet:
        print(r.stdout[-2000:])
    pin_libm()


PINNED = os.path.join(HERE, "_ref", "pinned_libm")


def pin_libm():
    """oracle/_ref/pinned_libm/libm.so.6 = the libm of the container that BUILT the reference, + its glibc version.  The arbiter of
    the full-size test (tests/test_gpu_dropin.py) is Mitsuba with glibc's <= 1-ulp transcendentals; its subprocess puts this directory first on
    LD_LIBRARY_PATH, so that the comparison runs against the SAME libm on every GPU lease (the directory travels with the snapshot like the other built
    files).  Static linking is not an option: glibc's libm.a is not position independent."""
    import ctypes
    import shutil
    src = None
    for line in open("/proc/self/maps"):
        if "/libm.so" in line or "/libm-" in line:
            src = line.split()[-1]
            break
    if src is None:
        ctypes.CDLL("libm.so.6")
        for line in open("/proc/self/maps"):
            if "/libm.so" in line or "/libm-" in line:
                src = line.split()[-1]
                break
    if src is None:
        return
    os.makedirs(PINNED, exist_ok=True)
    shutil.copy2(os.path.realpath(src), os.path.join(PINNED, "libm.so.6"))
    ver = ctypes.CDLL(None).gnu_get_libc_version
    ver.restype = ctypes.c_char_p
    open(os.path.join(PINNED, "glibc_version.txt"), "w").write(ver().decode() + "\n")


def build_shims():
    """oracle/_ref/plugins/{path_hip,direct_hip}.so: the product's Mitsuba plugins compiled against the reference (needs
    libphip.so and the reference tree)"""
    r = subprocess.run(["make", "-C", HERE, "-f", "Makefile.ref", "shims", "REF=" + REFERENCE], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("shim build failed:\n" + r.stdout[-3000:] + r.stderr[-3000:])


def have_shims():
    return all(os.path.exists(os.path.join(HERE, "_ref", "plugins", n + ".so")) for n in ("path_hip", "direct_hip", "volpath_simple_hip"))


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB):
        build()
    L = C.CDLL(LIB)
    fp, u8p, u32 = C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.c_uint32
    L.ref_last_error.restype = C.c_char_p
    L.ref_scene_create.restype = C.c_void_p
    L.ref_scene_create.argtypes = [C.POINTER(A.phip_scene_desc), C.c_float]
    L.ref_scene_destroy.argtypes = [C.c_void_p]
    L.ref_render.argtypes = [C.c_void_p, C.POINTER(A.phip_render_params), fp, fp]
    L.ref_render_job.argtypes = [C.c_void_p, C.POINTER(A.phip_render_params), C.c_int, fp, C.POINTER(C.c_double)]
    L.ref_render_job_plugin.argtypes = [C.c_void_p, C.POINTER(A.phip_render_params), C.c_char_p, C.c_int, fp, C.POINTER(C.c_double)]
    L.ref_trace.argtypes = [C.c_void_p, fp, C.c_size_t, fp]
    L.ref_intersect.argtypes = [C.c_void_p, fp, C.c_size_t, fp]
    L.ref_bsdf_sample.argtypes = [C.c_void_p, u32, C.c_size_t, fp, fp, fp, fp, fp, u8p]
    L.ref_bsdf_eval_pdf.argtypes = [C.c_void_p, u32, C.c_size_t, fp, fp, fp, fp]
    L.ref_sample_emitter.argtypes = [C.c_void_p, fp, fp, C.c_size_t, fp, fp, fp, fp, fp, fp]
    L.ref_camera_ray.argtypes = [C.c_void_p, C.c_float, C.c_float, fp]
    L.ref_set_sampler.argtypes = [C.c_int]
    L.ref_set_analytic_rectangles.argtypes = [C.c_int]
    L.ref_add_shape_file.argtypes = [C.c_char_p, C.c_char_p, u32, fp]
    L.ref_write_serialized.argtypes = [C.POINTER(A.phip_scene_desc), u32, C.c_char_p]
    L.ref_qmc_tables.argtypes = [C.c_int, u32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint16)]
    L.ref_mip_build.restype = C.c_void_p
    L.ref_mip_build.argtypes = [C.c_int, fp, u32, u32, u32, u32, u32, C.c_float]
    L.ref_mip_levels.argtypes = [C.c_void_p]
    L.ref_mip_level_size.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ref_mip_level_data.argtypes = [C.c_void_p, C.c_int, fp]
    L.ref_mip_eval.argtypes = [C.c_void_p, C.c_size_t, fp, fp, fp, fp]
    L.ref_mip_destroy.argtypes = [C.c_void_p]
    if L.ref_init() != 0:
        raise RuntimeError("ref_init: " + L.ref_last_error().decode())
    import atexit
    atexit.register(L.ref_shutdown)
    _lib = L
    return L


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class RefScene:
    """The reference's Scene object built from a phip_scene_desc (environment emitters must be listed first)."""

    def __init__(self, desc, stddev=0.5, analytic_rectangles=False, shape_files=()):
        """analytic_rectangles: exact rectangles of the description become the reference's analytic `rectangle` shape (its
        own intersection / sampling code on the CPU; the plugin shims see them through Shape::createTriMesh)"""
        self.L = lib()
        self.desc = desc
        self.L.ref_set_analytic_rectangles(1 if analytic_rectangles else 0)
        # shape_files: (plugin, filename, material id[, row-major 4x4 toWorld]) loaded by the reference's own mesh-loader plugins
        for sf in shape_files:
            tw = np.ascontiguousarray(sf[3] if len(sf) > 3 else np.eye(4), np.float32)
            self.L.ref_add_shape_file(sf[0].encode(), sf[1].encode(), int(sf[2]), tw.ctypes.data_as(C.POINTER(C.c_float)))
        self.h = self.L.ref_scene_create(C.byref(desc), stddev)
        self.L.ref_set_analytic_rectangles(0)
        if not self.h:
            raise RuntimeError("ref_scene_create: " + self.L.ref_last_error().decode())
        self.width, self.height = desc.film.crop_width, desc.film.crop_height

    def close(self):
        if self.h:
            self.L.ref_scene_destroy(self.h)
            self.h = None

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(what + ": " + self.L.ref_last_error().decode())

    def _sampler(self, sampler):
        self.L.ref_set_sampler({"independent": 0, "ctr": 1, "ldsampler": 2, "sobol": 3, "stratified": 4, "halton": 5, "hammersley": 6}[sampler])

    def render(self, params, want_samples=True, sampler="independent"):
        """sampler="ctr": the reference's integrator fed with the counter-based parity stream (ref_glue/ctr_sampler.cpp: defined by
        call order, so the plugin needs to know nothing about the scene)."""
        self._sampler(sampler)
        film = np.zeros((self.height, self.width, 5), np.float32)
        samples = np.zeros((self.height, self.width, params.spp, 4), np.float32) if want_samples else None
        self._check(self.L.ref_render(self.h, C.byref(params), _fp(samples) if want_samples else None, _fp(film)), "ref_render")
        return film, samples

    def render_job(self, params, threads=None, want_image=True, plugin=None, sampler="independent"):
        """the reference's complete multi-threaded render (RenderJob on the Scheduler); returns (rgb or None, seconds).
        plugin="path_hip" / "direct_hip": the same job with the product's plugin shim as the scene's integrator."""
        threads = threads or os.cpu_count() or 1
        self._sampler(sampler)
        rgb = np.zeros((self.height, self.width, 3), np.float32) if want_image else None
        sec = C.c_double()
        self._check(self.L.ref_render_job_plugin(self.h, C.byref(params), plugin.encode() if plugin else None, threads,
                                                 _fp(rgb) if want_image else None, C.byref(sec)), "ref_render_job")
        return rgb, sec.value

    def trace(self, rays):
        r = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        out = np.zeros((len(r), 5), np.float32)
        self._check(self.L.ref_trace(self.h, _fp(r), len(r), _fp(out)), "ref_trace")
        return out

    def intersect(self, rays):
        r = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        out = np.zeros((len(r), 20), np.float32)
        self._check(self.L.ref_intersect(self.h, _fp(r), len(r), _fp(out)), "ref_intersect")
        return out

    def bsdf_sample(self, material, wi, samples):
        wi = np.ascontiguousarray(wi, np.float32).reshape(-1, 3); s = np.ascontiguousarray(samples, np.float32).reshape(-1, 2)
        n = len(s)
        wo = np.zeros((n, 3), np.float32); w = np.zeros((n, 3), np.float32); pdf = np.zeros(n, np.float32); delta = np.zeros(n, np.uint8)
        self._check(self.L.ref_bsdf_sample(self.h, material, n, _fp(wi), _fp(s), _fp(wo), _fp(w), _fp(pdf),
                                           delta.ctypes.data_as(C.POINTER(C.c_uint8))), "ref_bsdf_sample")
        return wo, w, pdf, delta

    def bsdf_eval_pdf(self, material, wi, wo):
        wi = np.ascontiguousarray(wi, np.float32).reshape(-1, 3); wo = np.ascontiguousarray(wo, np.float32).reshape(-1, 3)
        n = len(wo)
        v = np.zeros((n, 3), np.float32); pdf = np.zeros(n, np.float32)
        self._check(self.L.ref_bsdf_eval_pdf(self.h, material, n, _fp(wi), _fp(wo), _fp(v), _fp(pdf)), "ref_bsdf_eval_pdf")
        return v, pdf

    def sample_emitter(self, ref, refN, samples):
        s = np.ascontiguousarray(samples, np.float32).reshape(-1, 2)
        n = len(s)
        ref = np.ascontiguousarray(ref, np.float32); refN = np.ascontiguousarray(refN, np.float32)
        d = np.zeros((n, 3), np.float32); dist = np.zeros(n, np.float32); pdf = np.zeros(n, np.float32)
        val = np.zeros((n, 3), np.float32); chk = np.zeros(n, np.float32)
        self._check(self.L.ref_sample_emitter(self.h, _fp(ref), _fp(refN), n, _fp(s), _fp(d), _fp(dist), _fp(pdf), _fp(val), _fp(chk)), "ref_sample_emitter")
        return d, dist, pdf, val, chk

    def camera_ray(self, sx, sy):
        out = np.zeros(14, np.float32)
        self._check(self.L.ref_camera_ray(self.h, sx, sy, _fp(out)), "ref_camera_ray")
        return out


WRAP = {"clamp": A.PHIP_WRAP_CLAMP, "repeat": A.PHIP_WRAP_REPEAT, "mirror": A.PHIP_WRAP_MIRROR, "zero": A.PHIP_WRAP_ZERO, "one": A.PHIP_WRAP_ONE}
FILTER = {"nearest": 0, "bilinear": 1, "trilinear": 2, "ewa": 3}


class RefMip:
    """TMIPMap built by the reference's own code (render/mipmap.h) the way envmap.cpp (kind="envmap") or bitmap.cpp
    (kind="texture") does: Lanczos-resampled levels, stored in half precision."""

    def __init__(self, image, kind="texture", wrap_u="repeat", wrap_v=None, filter_type="ewa", max_anisotropy=None):
        self.L = lib()
        img = np.ascontiguousarray(image, np.float32)
        h, w = img.shape[:2]
        if max_anisotropy is None:
            max_anisotropy = 10.0 if kind == "envmap" else 20.0
        self.h = self.L.ref_mip_build(0 if kind == "envmap" else 1, _fp(img), w, h, WRAP[wrap_u], WRAP[wrap_v or wrap_u],
                                      FILTER[filter_type], max_anisotropy)
        if not self.h:
            raise RuntimeError("ref_mip_build: " + self.L.ref_last_error().decode())
        self.levels = []
        for l in range(self.L.ref_mip_levels(self.h)):
            lw, lh = C.c_int(), C.c_int()
            self.L.ref_mip_level_size(self.h, l, C.byref(lw), C.byref(lh))
            a = np.zeros((lh.value, lw.value, 3), np.float32)
            self.L.ref_mip_level_data(self.h, l, _fp(a))
            self.levels.append(a)

    def eval(self, uv, d0, d1):
        uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2); d0 = np.ascontiguousarray(d0, np.float32).reshape(-1, 2)
        d1 = np.ascontiguousarray(d1, np.float32).reshape(-1, 2)
        out = np.zeros((len(uv), 3), np.float32)
        if self.L.ref_mip_eval(self.h, len(uv), _fp(uv), _fp(d0), _fp(d1), _fp(out)) != 0:
            raise RuntimeError("ref_mip_eval: " + self.L.ref_last_error().decode())
        return out

    def close(self):
        if self.h:
            self.L.ref_mip_destroy(self.h)
            self.h = None


def sobol_tables(width, height, dimensions=1024):
    """The direction numbers of the reference's OWN sobol plugin (oracle/_ref/plugins/sobol.so = src/samplers/sobol.cpp + sobolseq.cpp,
    compiled in place), read out of the loaded library as data: (matrices32[dimensions * 52], vdc[52], vdc_inv[52], log_resolution) for a
    film whose crop window is width x height (SobolSampler::setFilmResolution, sobol.cpp:147-157).  Nothing is copied into the repository."""
    so = C.CDLL(os.path.join(HERE, "_ref", "plugins", "sobol.so"))
    res, m = 1, 0
    while res < max(width, height):
        res <<= 1; m += 1
    mat = (C.c_uint32 * (1024 * 52)).in_dll(so, "_ZN5sobol8Matrices10matrices32E")
    matrices = np.frombuffer(mat, dtype=np.uint32)[:dimensions * 52].copy()
    vdc = np.zeros(52, np.uint64); inv = np.zeros(52, np.uint64)
    if m > 1:
        rows = 26                                           # vdc_sobol_matrices[m - 1], m = 1 .. 26 (sobolseq.cpp:106537, 107241)
        a = np.frombuffer((C.c_uint64 * (rows * 52)).in_dll(so, "_ZN5sobol8Matrices18vdc_sobol_matricesE"), dtype=np.uint64).reshape(rows, 52)
        b = np.frombuffer((C.c_uint64 * (rows * 52)).in_dll(so, "_ZN5sobol8Matrices22vdc_sobol_matrices_invE"), dtype=np.uint64).reshape(rows, 52)
        vdc, inv = a[m - 1].copy(), b[m - 1].copy()
    return matrices, vdc, inv, m


def sobol_scramble(frame):
    """phip_render_params.sobol_scramble for the plugin's `scramble` property (a frame number): 0 stays 0, anything else goes through four
    rounds of TEA (sobol.cpp:92-102 -> qmc.h:146-156, Wheeler & Needham's block cipher)"""
    if frame == 0:
        return 0
    v0, v1, s, M = frame & 0xFFFFFFFF, (frame >> 32) & 0xFFFFFFFF, 0, 0xFFFFFFFF
    for _ in range(4):
        s = (s + 0x9e3779b9) & M
        v0 = (v0 + ((((v1 << 4) & M) + 0xA341316C) ^ (v1 + s) ^ ((v1 >> 5) + 0xC8013EA4))) & M
        v1 = (v1 + ((((v0 << 4) & M) + 0xAD90777D) ^ (v0 + s) ^ ((v0 >> 5) + 0x7E95761E))) & M
    return (v1 << 32) + v0


def write_serialized(desc, shape, path):
    """shape `shape` of a scene description as a .serialized mesh file, written by the reference's own TriMesh::serialize"""
    if lib().ref_write_serialized(C.byref(desc), int(shape), path.encode()) != 0:
        raise RuntimeError("ref_write_serialized: " + lib().ref_last_error().decode())


def qmc_tables(scramble=-1, dimensions=128):
    """(primes[dimensions], permutations or None) for `default_render_params(qmc=...)`: the reference's prime table and the digit permutations its
    PermutationStorage builds for `scramble` (-1: Faure's, the samplers' default; 0: none; otherwise pseudorandom ones), read out of the reference"""
    primes = np.zeros(dimensions, np.uint32)
    if lib().ref_qmc_tables(0, dimensions, primes.ctypes.data_as(C.POINTER(C.c_uint32)), None) != 0:
        raise RuntimeError(lib().ref_last_error().decode())
    if scramble == 0:
        return primes, None
    perm = np.zeros(int(primes.sum()), np.uint16)
    if lib().ref_qmc_tables(int(scramble), dimensions, primes.ctypes.data_as(C.POINTER(C.c_uint32)), perm.ctypes.data_as(C.POINTER(C.c_uint16))) != 0:
        raise RuntimeError(lib().ref_last_error().decode())
    return primes, perm
