"""ctypes binding of libphip.so (include/phip.h) -- the only way the Python harness reaches the GPU.

The library is built in-tree by `build()` (hipcc --offload-arch=gfx950) into
mitsuba_amd/_build/libphip.so; there is no fallback: if it is missing or cannot be loaded the
import of anything that renders raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import _abi as A

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(HERE, "_build")
LIB = os.path.join(BUILD, "libphip.so")
# the same sources with the test hooks compiled in (phip_debug.inl: host twins of device functions, the fmath probe, the HBM calibration kernels): what tests/ and tools/ reach
# through lib().phip_debug_*; the product library does not export them
LIB_DEBUG = os.path.join(BUILD, "libphip_debug.so")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-Wall", "-Wno-unused-function"]

_lib = None


def _sources():
    """the files a build depends on: *.hip / *.h / *.inl of csrc/ and include/ (an editor backup must not mark the library stale)"""
    keep = (".hip", ".h", ".inl")
    out = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(keep)]
    inc = os.path.join(os.path.dirname(HERE), "include")
    out += [os.path.join(inc, f) for f in sorted(os.listdir(inc)) if f.endswith(keep)]
    return out


# objects of libphip.so: (source, extra flags, object name) -- see csrc/phip_common.h
# phip_mega.hip is compiled without MachineLICM: hoisting the double-precision polynomial constants of phip_fmath.h (two VGPRs each,
# 64-bit literals cannot be encoded) out of k_mega's persistent loop cost ~40 VGPRs -- 168 instead of 128, i.e. 3 instead of 4 waves per SIMD
MEGA_FLAGS = ["-mllvm", "-disable-machine-licm"]
# phip_shade.hip: once per feature set and part (see its header): 24 objects; the heavy ones (textures: bit 1 of the feature set) first, so that the
# longest compiles start when the pool does
SHADE_FEATS, SHADE_PARTS = (11, 3, 2, 1, 8, 0), (0, 1, 3, 2)
UNITS = [("phip_shade.hip", ["-DSHADE_FEAT=%d" % f, "-DSHADE_PART=%d" % q], "phip_shade%d_%d.o" % (f, q)) for f in SHADE_FEATS for q in SHADE_PARTS] + \
        [("phip_mega.hip", MEGA_FLAGS + ["-DMEGA_PART=0"], "phip_mega.o"), ("phip_mega.hip", MEGA_FLAGS + ["-DMEGA_PART=1"], "phip_megaw.o"), ("phip_mega.hip", MEGA_FLAGS + ["-DMEGA_PART=2"], "phip_megad.o"), ("phip.hip", [], "phip.o")]


DEBUG_UNIT = ("phip.hip", ["-DPHIP_DEBUG_HOOKS=1"], "phip_dbg.o")


def source_id():
    """Provenance of a build: a hash of every source file and the compile flags.  It is compiled into the library (phip_build_id) so
    that a stale libphip.so -- file times mean nothing after a copy to another machine -- is recognised when it is loaded."""
    import hashlib
    h = hashlib.sha256()
    for s in _sources():
        h.update(os.path.basename(s).encode()); h.update(open(s, "rb").read())
    h.update(repr((HIPCC_FLAGS, UNITS, DEBUG_UNIT, os.environ.get("PHIP_EXTRA_HIPCC_FLAGS", ""))).encode())
    return h.hexdigest()[:16]


def built_id(path=None):
    """the source id compiled into an existing library (None: no library / built before ids existed)"""
    path = path or LIB
    if not os.path.exists(path):
        return None
    data = open(path, "rb").read()
    i = data.find(b"phip-build-id:")
    return data[i + 14:i + 30].decode() if i >= 0 else None


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 (cross-compiles without a GPU): the objects in parallel, then one link."""
    os.makedirs(BUILD, exist_ok=True)
    sid = source_id()
    out_lib = os.environ.get("PHIP_BUILD_OUTPUT", LIB)      # experiment hook: alternative builds next to the product (load with PHIP_LIB)
    alt = os.path.abspath(out_lib) != os.path.abspath(LIB)          # an alternative build (fault injection, small stacks): the product's sources with other flags, no test hooks
    if not force and built_id(out_lib) == sid and (alt or built_id(LIB_DEBUG) == sid):
        return out_lib
    # several processes may find the library stale at once (ranks of a torchrun job, pytest-xdist workers): one builds, the others
    # wait for the lock and find the fresh file; objects and library are written under temporary names and renamed into place
    import fcntl
    with open(os.path.join(BUILD, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and built_id(out_lib) == sid and (alt or built_id(LIB_DEBUG) == sid):
            return out_lib
        return _build_locked(sid, out_lib, verbose)


def _build_locked(sid, out_lib, verbose):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = [f for f in HIPCC_FLAGS if f != "-shared"] + os.environ.get("PHIP_EXTRA_HIPCC_FLAGS", "").split()
    cmds = []
    # an alternative build (PHIP_BUILD_OUTPUT) keeps objects of its own: tools/build_variant.sh links the PRODUCT's objects by name
    sfx = "" if os.path.abspath(out_lib) == os.path.abspath(LIB) else "-" + os.path.splitext(os.path.basename(out_lib))[0]
    units = [(src, extra, obj[:-2] + sfx + ".o") for src, extra, obj in UNITS]
    # PHIP_BUILD_REUSE="phip_shade.hip ...": an alternative build whose extra flags do not concern these sources links the PRODUCT's objects of them (where they exist
    # and are the current build's: this container; on the GPU box objects do not travel and everything is compiled)
    reuse = set(os.environ.get("PHIP_BUILD_REUSE", "").split()) if sfx and built_id(LIB) == sid else set()
    reused = set()
    for i, (src, extra, obj) in enumerate(units):
        prod = UNITS[i][2]
        if src in reuse and os.path.exists(os.path.join(BUILD, prod)) and os.path.getmtime(os.path.join(BUILD, prod)) <= os.path.getmtime(LIB):
            units[i] = (src, extra, prod); reused.add(prod)
    dbg = None if sfx else DEBUG_UNIT                           # (the hooks: the product build only)
    for src, extra, obj in units + ([dbg] if dbg else []):
        if obj in reused:
            continue
        if src == "phip.hip":
            extra = extra + ['-DPHIP_BUILD_ID="%s"' % sid]
        cmds.append([hipcc] + flags + extra + ["-c", os.path.join(CSRC, src), "-o", os.path.join(BUILD, obj + ".tmp.o")])
    # as many compilers at a time as there are cores, longest unit first (measured seconds per feature set and part; the order of UNITS is part of the build id,
    # the order of submission is not)
    feat_cost = {11: 55.0, 3: 41.0, 2: 29.0, 1: 14.0, 8: 9.0, 0: 11.0}
    def cost(cmd):
        name = os.path.basename(cmd[-1])
        if name.startswith("phip_shade"):
            import re
            f, q = re.match(r"phip_shade(\d+)_(\d)", name).groups()
            return feat_cost.get(int(f), 20.0) * (0.42 if q == "2" else 1.0)
        return 31.0 if name.startswith("phip_mega") else 14.0
    cmds.sort(key=cost, reverse=True)
    import concurrent.futures
    def run(cmd):
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        return cmd, r.returncode, r.stdout
    with concurrent.futures.ThreadPoolExecutor(max_workers=max(2, os.cpu_count() or 4)) as pool:
        for cmd, rc, out in pool.map(run, cmds):
            if rc != 0:
                raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + out)
            if verbose:
                print(" ".join(cmd)); print(out)
    for _, _, obj in units + ([dbg] if dbg else []):
        if obj not in reused:
            os.replace(os.path.join(BUILD, obj + ".tmp.o"), os.path.join(BUILD, obj))
    links = [(out_lib, [u[2] for u in units])]
    if dbg:
        links.append((LIB_DEBUG, [dbg[2] if u[0] == "phip.hip" else u[2] for u in units]))
    for target, objs in links:
        tmp = target + ".tmp.%d" % os.getpid()
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + [os.path.join(BUILD, o) for o in objs] + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        os.replace(tmp, target)                             # a concurrent loader sees the old library or the new one, never half of one
    return out_lib


# The libraries of two GPU tests (tests/test_gpu_parity.py): the product's sources with a fault or a limit compiled in.  __graft_entry__.build() makes them in the build
# container so that they travel with the snapshot (a whole-library build on the GPU box costs the suite two minutes each); the tests call the same function, which
# returns at once when the file carries the current build id.
TEST_VARIANTS = {"fault": "-DMEGA_MB_FAULT=1",       # the first wave of every fused launch reports that it gave up: the frame must come from the re-rendered pass
                 "cap32": "-DWP_CAP=32u"}            # 32-entry task stacks: nearly every push of a big scene spills to memory


def build_test_variant(tag):
    """mitsuba_amd/_build/libphip_<tag>.so (load it with PHIP_LIB); raises with the compiler's output when the build fails"""
    import sys
    out = os.path.join(BUILD, "libphip_%s.so" % tag)
    root = os.path.dirname(HERE)
    env = dict(os.environ, PHIP_BUILD_OUTPUT=out, PHIP_EXTRA_HIPCC_FLAGS=TEST_VARIANTS[tag], PHIP_BUILD_REUSE="phip_shade.hip")
    env.pop("PHIP_LIB", None)
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from mitsuba_amd import _ffi; print(_ffi.build())" % root], env=env, capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(out):
        raise RuntimeError("building %s failed:\n%s%s" % (out, r.stdout[-2000:], r.stderr[-2000:]))
    return out


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("PHIP_LIB", LIB)     # experiment hook: alternative builds of the same sources
    if not os.path.exists(path):
        raise RuntimeError("libphip.so is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "-- path_hip has no CPU fallback" % LIB)
    if "PHIP_LIB" not in os.environ and built_id(path) != source_id() and os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        build()                                 # stale (or missing id): rebuild from the sources next to it
    if "PHIP_LIB" not in os.environ and built_id(path) != source_id():
        raise RuntimeError("libphip.so (%s, build id %s) was not built from the sources next to it (id %s): rebuild with "
                           "`python -c 'import __graft_entry__ as g; g.build()'`" % (path, built_id(path), source_id()))
    L = C.CDLL(path)
    fp, u8p, u32 = C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.c_uint32
    L.phip_build_id.restype = C.c_char_p
    L.phip_last_error.restype = C.c_char_p
    L.phip_version.restype = C.c_char_p
    L.phip_device_count.restype = C.c_int
    L.phip_scene_create.restype = C.c_void_p
    L.phip_scene_create.argtypes = [C.POINTER(A.phip_scene_desc), C.c_int]
    L.phip_scene_destroy.argtypes = [C.c_void_p]
    L.phip_render.argtypes = [C.c_void_p, C.POINTER(A.phip_render_params), fp, C.POINTER(A.phip_stats)]
    L.phip_render_device.argtypes = [C.c_void_p, C.POINTER(A.phip_render_params), C.c_void_p, C.POINTER(A.phip_stats)]
    L.phip_get_samples.argtypes = [C.c_void_p, fp, C.c_size_t]
    L.phip_trace.argtypes = [C.c_void_p, C.POINTER(A.phip_ray), C.c_size_t, C.POINTER(A.phip_hit), u8p, C.POINTER(A.phip_stats)]
    L.phip_cancel.argtypes = [C.c_void_p]
    L.phip_scene_replicate.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int32]
    L.phip_develop.argtypes = [fp, C.c_size_t, fp]
    L.phip_scene_accel_info.argtypes = [C.c_void_p, C.POINTER(A.phip_accel_info)]
    L.phip_gaussian_filter.argtypes = [C.c_float, fp, fp]
    L.phip_film_to_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.phip_host_alloc.restype = C.c_void_p
    L.phip_host_alloc.argtypes = [C.c_size_t]
    L.phip_host_free.argtypes = [C.c_void_p]; L.phip_host_free.restype = None
    L.phip_abi_sizeof.restype = C.c_size_t
    L.phip_abi_sizeof.argtypes = [C.c_int]
    _lib = _LibWithHooks(L)
    return _lib


class _LibWithHooks(object):
    """the product library; attribute look-ups of the test hooks (phip_debug_*) go to libphip_debug.so, which is loaded at the first of them"""
    def __init__(self, product):
        object.__setattr__(self, "_product", product)

    def __getattr__(self, name):
        if name.startswith("phip_debug_"):
            return getattr(debug_lib(), name)
        return getattr(object.__getattribute__(self, "_product"), name)


_debug = None


def debug_lib():
    """mitsuba_amd/_build/libphip_debug.so: the product's sources + phip_debug.inl (tests / tools only)"""
    global _debug
    if _debug is not None:
        return _debug
    path = os.environ.get("PHIP_DEBUG_LIB", LIB_DEBUG)
    if not os.path.exists(path):
        raise RuntimeError("libphip_debug.so is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
    L = C.CDLL(path)
    fp, u8p, u32 = C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.c_uint32
    L.phip_last_error.restype = C.c_char_p
    L.phip_debug_host_bsdf_sample.argtypes = [C.POINTER(A.phip_material), u32, u32, C.c_size_t, fp, fp, fp, fp, fp, u8p]
    L.phip_debug_host_bsdf_eval_pdf.argtypes = [C.POINTER(A.phip_material), u32, u32, C.c_size_t, fp, fp, fp, fp]
    L.phip_debug_host_camera_ray.argtypes = [C.POINTER(A.phip_camera), C.POINTER(A.phip_film), C.c_float, C.c_float, C.POINTER(A.phip_ray)]
    L.phip_debug_host_ctr_block.argtypes = [u32, u32, u32, u32, fp]
    L.phip_debug_host_ld_point.argtypes = [u32, u32, u32, u32, u32, fp]; L.phip_debug_host_ld_point.restype = None
    u32p, u64p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    L.phip_debug_host_sobol.argtypes = [u32p, u32, u64p, u64p, u32, u32, C.c_int, C.c_size_t, u32p, u32p, u32p, u32p, u64p, fp, fp]
    L.phip_debug_host_rinv.argtypes = [u32p, C.POINTER(C.c_uint16), u32, C.c_int, C.c_size_t, u64p, u32p, fp]
    L.phip_debug_host_rinv2.argtypes = [u32p, C.POINTER(C.c_uint16), u32, C.c_int, C.c_size_t, u64p, u32p, fp]
    L.phip_debug_host_cdf_sample.argtypes = [fp, u32, fp, C.c_size_t, C.POINTER(C.c_uint32)]; L.phip_debug_host_cdf_sample.restype = None
    L.phip_debug_host_mip_eval.argtypes = [C.POINTER(A.phip_texture), C.c_size_t, fp, fp, fp, fp]
    L.phip_debug_host_build_bvh.argtypes = [fp, u32, C.POINTER(C.c_uint32), u32, C.POINTER(A.phip_accel_info), fp]
    L.phip_debug_host_trace_wide.argtypes = [fp, u32, C.POINTER(C.c_uint32), u32, C.POINTER(A.phip_ray), C.c_size_t, C.POINTER(A.phip_hit), C.c_int, C.POINTER(A.phip_accel_info), u8p, u32]
    L.phip_debug_fmath.argtypes = [C.c_int, C.c_int, C.c_size_t, fp, fp, fp]
    _debug = L
    return L


def last_error():
    return lib().phip_last_error().decode()


class PhipError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        super().__init__("%s failed (%d): %s" % (where, code, last_error()))


def gaussian_filter(stddev=0.5):
    """(radius, table[32]) of the reference's default `gaussian` rfilter."""
    r = C.c_float()
    t = (C.c_float * 32)()
    lib().phip_gaussian_filter(stddev, C.byref(r), t)
    return (r.value, list(t))


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))
