"""Host-side mirror of the reference's integrator plugin interface for the `path` hot path.

Names, parameter meaning and error behaviour follow the reference so that tests read like the
reference's own:

    PathHIP(props)           <-> MIPathTracer(const Properties &)           src/integrators/path/path.cpp:109-111
      maxDepth / rrDepth / strictNormals / hideEmitters                      src/librender/integrator.cpp:190-225
    PathHIP.render(scene)    <-> SamplingIntegrator::render(scene, queue, job, ...) -> bool   integrator.cpp:95-129
    PathHIP.cancel()         <-> SamplingIntegrator::cancel                  integrator.cpp:90-93
    DirectHIP(props)         <-> MIDirectIntegrator(const Properties &)      src/integrators/direct/direct.cpp:91-108
      shadingSamples / emitterSamples / bsdfSamples / strictNormals / hideEmitters
    Scene(desc)              <-> Scene::initialize (kd-tree build -> BVH build + upload)       scene.cpp:322-384
    HDRFilm.put / develop    <-> HDRFilm::put(const ImageBlock *) / develop  films/hdrfilm.cpp:391-393,427-475

Everything below the C ABI runs on the GPU; this module only marshals arguments.
"""
import ctypes as C

import numpy as np

from . import _abi as A
from . import _ffi


class Properties(dict):
    """Minimal typed property bag (include/mitsuba/core/properties.h)."""

    def __init__(self, plugin_name="", **kw):
        super().__init__(**kw)
        self.plugin_name = plugin_name

    def getInteger(self, name, default):
        return int(self.get(name, default))

    def getBoolean(self, name, default):
        return bool(self.get(name, default))

    def getSize(self, name, default):
        v = int(self.get(name, default))
        if v < 0:
            raise RuntimeError("Size property '%s': expected a nonnegative value!" % name)     # properties.cpp getSize
        return v


class Scene:
    """Device-resident scene: flattened meshes + BVH + emitter tables (phip_scene)."""

    def __init__(self, desc, device=0):
        self._L = _ffi.lib()
        self.desc = desc
        self.device = device
        self._h = self._L.phip_scene_create(C.byref(desc), device)
        if not self._h:
            raise RuntimeError("phip_scene_create: " + _ffi.last_error())
        self.width, self.height = desc.film.crop_width, desc.film.crop_height
        self.block_size = 32          # Scene::getBlockSize default (mitsuba.cpp:144)

    def setBlockSize(self, bs):
        self.block_size = int(bs)

    def accel_info(self):
        info = A.phip_accel_info()
        rc = self._L.phip_scene_accel_info(self._h, C.byref(info))
        if rc != 0:
            raise _ffi.PhipError(rc, "phip_scene_accel_info")
        return info

    def rayIntersect(self, rays, closest=True, shadow=False):
        """Batch version of ShapeKDTree::rayIntersect(ray, its) / (ray).  rays: (n,8) float32 = o, mint, d, maxt."""
        r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = len(r)
        hits = np.zeros((n, 4), np.float32) if closest else None
        occ = np.zeros(n, np.uint8) if shadow else None
        st = A.phip_stats()
        rc = self._L.phip_trace(self._h, r.ctypes.data_as(C.POINTER(A.phip_ray)), n,
                                hits.ctypes.data_as(C.POINTER(A.phip_hit)) if closest else None,
                                occ.ctypes.data_as(C.POINTER(C.c_uint8)) if shadow else None, C.byref(st))
        if rc != 0:
            raise _ffi.PhipError(rc, "phip_trace")
        return hits, occ, st

    def replicate(self, devices):
        """Copies the device scene to further GPUs ahead of a multi-device render (phip_scene_replicate)."""
        arr = (C.c_int32 * len(devices))(*devices)
        rc = self._L.phip_scene_replicate(self._h, arr, len(devices))
        if rc != 0:
            raise _ffi.PhipError(rc, "phip_scene_replicate")

    def film_to_host(self, d_ptr, host_ptr):
        """a device-resident frame (render_device's output, e.g. after an RCCL reduce) to host memory, as phip_render delivers it"""
        rc = self._L.phip_film_to_host(self._h, C.c_void_p(d_ptr), C.c_void_p(host_ptr))
        if rc != 0:
            raise _ffi.PhipError(rc, "phip_film_to_host")

    def close(self):
        if getattr(self, "_h", None):
            self._L.phip_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HDRFilm:
    """Crop-sized (R,G,B,alpha,weight) float32 accumulation buffer, like HDRFilm's m_storage."""

    def __init__(self, width, height):
        self.storage = np.zeros((height, width, 5), np.float32)

    def clear(self):
        self.storage[...] = 0

    def put(self, block):
        self.storage += block

    def develop(self):
        h, w, _ = self.storage.shape
        out = np.zeros((h, w, 3), np.float32)
        _ffi.lib().phip_develop(_ffi.fptr(self.storage), h * w, _ffi.fptr(out))
        return out


class PinnedFilm(HDRFilm):
    """HDRFilm whose storage is page-locked (phip_host_alloc): PathHIP.render_into(scene, film.ptr, spp) receives the frame in one
    asynchronous device-to-host copy instead of the staged copy pageable memory needs."""

    def __init__(self, width, height):
        n = width * height * 5
        self._p = _ffi.lib().phip_host_alloc(n * 4)
        if not self._p:
            raise RuntimeError("phip_host_alloc: " + _ffi.last_error())
        self.ptr = self._p
        self.storage = np.ctypeslib.as_array(C.cast(C.c_void_p(self._p), C.POINTER(C.c_float)), shape=(height, width, 5))
        self.storage[...] = 0

    def close(self):
        if getattr(self, "_p", None):
            self.storage = None
            _ffi.lib().phip_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PathHIP:
    """`path_hip` integrator: MIPathTracer semantics, MI355X execution."""

    def __init__(self, props=None, **kw):
        props = Properties("path_hip", **(dict(props or {}) | kw))
        self.m_rrDepth = props.getInteger("rrDepth", 5)
        self.m_maxDepth = props.getInteger("maxDepth", -1)
        self.m_strictNormals = props.getBoolean("strictNormals", False)
        self.m_hideEmitters = props.getBoolean("hideEmitters", False)
        # integrator.cpp:219-224
        if self.m_rrDepth <= 0:
            raise RuntimeError("'rrDepth' must be set to a value greater than zero!")
        if self.m_maxDepth <= 0 and self.m_maxDepth != -1:
            raise RuntimeError("'maxDepth' must be set to -1 (infinite) or a value greater than zero!")
        self.stats = None
        self._scene = None

    def params(self, scene, spp, seed=0, shard_index=0, shard_count=1, flags=0, stream=None, **extra):
        """extra: devices=[...] (multi-GPU inside the call), sample_offset / sample_total (progressive passes), progress=callable"""
        return A.default_render_params(spp=spp, max_depth=self.m_maxDepth, rr_depth=self.m_rrDepth,
                                       strict_normals=int(self.m_strictNormals), hide_emitters=int(self.m_hideEmitters),
                                       block_size=scene.block_size, seed=seed, shard_index=shard_index,
                                       shard_count=shard_count, device=scene.device, flags=flags, stream=stream, **extra)

    def render(self, scene, film, spp, seed=0, shard_index=0, shard_count=1, flags=0, **extra):
        """Renders into `film` (film.put of one full-frame block).  Returns True on success,
        False if cancelled (SamplingIntegrator::render returns proc->getReturnStatus() == ESuccess)."""
        self._scene = scene
        p = self.params(scene, spp, seed, shard_index, shard_count, flags, **extra)
        block = np.zeros((scene.height, scene.width, 5), np.float32)
        st = A.phip_stats()
        rc = _ffi.lib().phip_render(scene._h, C.byref(p), _ffi.fptr(block), C.byref(st))
        self.stats = st
        if rc == A.PHIP_ERR_CANCELLED:
            return False
        if rc != 0:
            raise _ffi.PhipError(rc, "phip_render")     # Log(EError, ...) throws in the reference
        film.put(block)
        return True

    def render_into(self, scene, host_ptr, spp, seed=0, shard_index=0, shard_count=1, flags=0, **extra):
        """phip_render into the caller's host memory (height x width x 5 float32 at address `host_ptr`): what the Mitsuba shim does with
        the film's bitmap.  Pinned memory (PinnedFilm below, torch pin_memory) gets the film in one asynchronous copy."""
        self._scene = scene
        p = self.params(scene, spp, seed, shard_index, shard_count, flags, **extra)
        st = A.phip_stats()
        rc = _ffi.lib().phip_render(scene._h, C.byref(p), C.cast(C.c_void_p(host_ptr), C.POINTER(C.c_float)), C.byref(st))
        self.stats = st
        if rc == A.PHIP_ERR_CANCELLED:
            return False
        if rc != 0:
            raise _ffi.PhipError(rc, "phip_render")
        return True

    def render_device(self, scene, d_out_ptr, spp, seed=0, shard_index=0, shard_count=1, flags=0, stream=None, **extra):
        """Renders this shard's blocks into device memory (e.g. a torch tensor) for an RCCL reduce."""
        self._scene = scene
        p = self.params(scene, spp, seed, shard_index, shard_count, flags, stream, **extra)
        st = A.phip_stats()
        rc = _ffi.lib().phip_render_device(scene._h, C.byref(p), C.c_void_p(d_out_ptr), C.byref(st))
        self.stats = st
        if rc == A.PHIP_ERR_CANCELLED:
            return False
        if rc != 0:
            raise _ffi.PhipError(rc, "phip_render_device")
        return True

    def samples(self, scene, spp):
        """Per-sample (R,G,B,alpha) of the last render made with PHIP_FLAG_SAMPLE_BUFFER: [y][x][sample]."""
        out = np.zeros((scene.height, scene.width, spp, 4), np.float32)
        rc = _ffi.lib().phip_get_samples(scene._h, _ffi.fptr(out), scene.height * scene.width * spp)
        if rc != 0:
            raise _ffi.PhipError(rc, "phip_get_samples")
        return out

    def cancel(self):
        if self._scene is not None:
            _ffi.lib().phip_cancel(self._scene._h)


class DirectHIP(PathHIP):
    """`direct_hip` integrator: MIDirectIntegrator semantics (direct.cpp:149-312) on the same kernels."""

    def __init__(self, props=None, **kw):
        props = Properties("direct_hip", **(dict(props or {}) | kw))
        # direct.cpp:94-107
        shadingSamples = props.getSize("shadingSamples", 1)
        self.m_emitterSamples = props.getSize("emitterSamples", shadingSamples)
        self.m_bsdfSamples = props.getSize("bsdfSamples", shadingSamples)
        self.m_strictNormals = props.getBoolean("strictNormals", False)
        self.m_hideEmitters = props.getBoolean("hideEmitters", False)
        if self.m_emitterSamples + self.m_bsdfSamples <= 0:
            raise RuntimeError("Assertion 'm_emitterSamples + m_bsdfSamples > 0' failed")
        self.m_rrDepth, self.m_maxDepth = 5, -1          # not parameters of this integrator
        self.stats = None
        self._scene = None

    def params(self, scene, spp, seed=0, shard_index=0, shard_count=1, flags=0, stream=None, **extra):
        p = super().params(scene, spp, seed, shard_index, shard_count, flags, stream, **extra)
        p.integrator = A.PHIP_INTEGRATOR_DIRECT
        p.emitter_samples = self.m_emitterSamples
        p.bsdf_samples = self.m_bsdfSamples
        return p


class VolPathSimpleHIP(PathHIP):
    """`volpath_simple_hip` integrator: SimpleVolumetricPathTracer (src/integrators/path/volpath_simple.cpp:88-318) on a scene without participating media --
    the `path` loop without multiple importance sampling, same parameters (MonteCarloIntegrator: maxDepth, rrDepth, strictNormals, hideEmitters).  The scene
    description has no media; the Mitsuba plugin shim (mitsuba_amd/plugin/volpath_simple_hip.cpp) refuses a scene that has any."""

    def params(self, scene, spp, seed=0, shard_index=0, shard_count=1, flags=0, stream=None, **extra):
        p = super().params(scene, spp, seed, shard_index, shard_count, flags, stream, **extra)
        p.integrator = A.PHIP_INTEGRATOR_VOLPATH_SIMPLE
        return p
