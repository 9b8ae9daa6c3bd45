"""ctypes mirror of include/phip.h (the C ABI of libphip.so).

Field order and types must match the header exactly; tests/test_abi.py checks sizeof() of every
struct against the values the C side reports through phip_abi_sizeof().
"""
import ctypes as C

PHIP_ABI_VERSION = 7
PHIP_FILTER_RESOLUTION = 31

PHIP_OK, PHIP_ERR_INVALID, PHIP_ERR_UNSUPPORTED, PHIP_ERR_DEVICE, PHIP_ERR_CANCELLED, PHIP_ERR_NOMEM = 0, -1, -2, -3, -4, -5
PHIP_BSDF_DIFFUSE, PHIP_BSDF_DIELECTRIC, PHIP_BSDF_ROUGHCONDUCTOR, PHIP_BSDF_TWOSIDED = 0, 1, 2, 3
PHIP_MF_BECKMANN, PHIP_MF_GGX = 0, 1
PHIP_SAMPLER_CTR = 0
PHIP_SAMPLER_LD = 1
PHIP_SAMPLER_SOBOL = 2
PHIP_SAMPLER_STRATIFIED = 3
PHIP_SAMPLER_HALTON = 4
PHIP_SAMPLER_HAMMERSLEY = 5
PHIP_SOBOL_MATRIX_SIZE = 52
PHIP_INTEGRATOR_PATH, PHIP_INTEGRATOR_DIRECT, PHIP_INTEGRATOR_VOLPATH_SIMPLE = 0, 1, 2
PHIP_FLAG_KERNEL_TIMING = 1
PHIP_FLAG_SAMPLE_BUFFER = 2
PHIP_FLAG_ENVMAP_BILINEAR_BACKGROUND = 4
PHIP_FLAG_ACCUMULATE = 8
PHIP_FLAG_ALIAS_DEVICES = 16
PHIP_FLAG_NO_FUSED = 32
PHIP_FLAG_NO_MEGA = 64
PHIP_FLAG_FUSED_ANY = 128
PHIP_FUSED_WIDE_MAX_NODES = 4096
PHIP_NO_HIT = 0xFFFFFFFF


class phip_material(C.Structure):
    _fields_ = [("type", C.c_uint32), ("nested", C.c_uint32 * 2),
                ("reflectance", C.c_float * 3), ("transmittance", C.c_float * 3),
                ("eta", C.c_float * 3), ("k", C.c_float * 3),
                ("alpha_u", C.c_float), ("alpha_v", C.c_float),
                ("distribution", C.c_uint32), ("sample_visible", C.c_uint32), ("reflectance_texture", C.c_uint32),
                ("alpha_u_texture", C.c_uint32), ("alpha_v_texture", C.c_uint32), ("transmittance_texture", C.c_uint32)]


class phip_shape(C.Structure):
    _fields_ = [("first_vertex", C.c_uint32), ("n_vertices", C.c_uint32),
                ("first_triangle", C.c_uint32), ("n_triangles", C.c_uint32),
                ("material", C.c_uint32), ("emitter", C.c_int32),
                ("has_normals", C.c_uint32), ("has_texcoords", C.c_uint32)]


PHIP_EMITTER_AREA, PHIP_EMITTER_CONSTANT, PHIP_EMITTER_ENVMAP = 0, 1, 2


class phip_emitter(C.Structure):
    _fields_ = [("radiance", C.c_float * 3), ("sampling_weight", C.c_float),
                ("shape", C.c_uint32), ("type", C.c_uint32), ("reserved", C.c_uint32 * 2)]


class phip_camera(C.Structure):
    _fields_ = [("to_world", C.c_float * 16), ("xfov_deg", C.c_float),
                ("near_clip", C.c_float), ("far_clip", C.c_float), ("reserved", C.c_float)]


class phip_film(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32),
                ("crop_offset_x", C.c_int32), ("crop_offset_y", C.c_int32),
                ("crop_width", C.c_int32), ("crop_height", C.c_int32),
                ("filter_radius", C.c_float),
                ("filter_table", C.c_float * (PHIP_FILTER_RESOLUTION + 1))]


PHIP_ENVMAP_MAX_LEVELS = 17


class phip_envmap(C.Structure):
    _fields_ = [("texels", C.POINTER(C.c_float)), ("width", C.c_uint32), ("height", C.c_uint32),
                ("scale", C.c_float), ("to_world", C.c_float * 16),
                ("n_levels", C.c_uint32), ("levels", C.POINTER(C.c_float) * PHIP_ENVMAP_MAX_LEVELS)]


PHIP_WRAP_CLAMP, PHIP_WRAP_REPEAT, PHIP_WRAP_MIRROR, PHIP_WRAP_ZERO, PHIP_WRAP_ONE = range(5)   # = EBoundaryCondition (rfilter.h:53-64)
PHIP_FILTER_NEAREST, PHIP_FILTER_BILINEAR, PHIP_FILTER_TRILINEAR, PHIP_FILTER_EWA = range(4)
PHIP_MIP_MAX_LEVELS = 17


class phip_texture(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("n_levels", C.c_uint32),
                ("levels", C.POINTER(C.c_float) * PHIP_MIP_MAX_LEVELS),
                ("wrap_u", C.c_uint32), ("wrap_v", C.c_uint32), ("filter_type", C.c_uint32),
                ("max_anisotropy", C.c_float), ("uv_scale", C.c_float * 2), ("uv_offset", C.c_float * 2)]


class phip_scene_desc(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("n_vertices", C.c_uint32),
                ("positions", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float)),
                ("n_triangles", C.c_uint32), ("indices", C.POINTER(C.c_uint32)),
                ("n_shapes", C.c_uint32), ("shapes", C.POINTER(phip_shape)),
                ("n_materials", C.c_uint32), ("materials", C.POINTER(phip_material)),
                ("n_emitters", C.c_uint32), ("emitters", C.POINTER(phip_emitter)),
                ("camera", phip_camera), ("film", phip_film), ("envmap", phip_envmap),
                ("texcoords", C.POINTER(C.c_float)), ("n_textures", C.c_uint32), ("textures", C.POINTER(phip_texture))]


PHIP_MAX_DEVICES = 16
# void (*progress)(void *user, int32_t device, uint64_t samples_done, uint64_t samples_total)
PROGRESS_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.c_uint64, C.c_uint64)


class phip_render_params(C.Structure):
    _fields_ = [("spp", C.c_int32), ("max_depth", C.c_int32), ("rr_depth", C.c_int32),
                ("strict_normals", C.c_int32), ("hide_emitters", C.c_int32), ("block_size", C.c_int32),
                ("sampler", C.c_uint32), ("seed", C.c_uint32),
                ("shard_index", C.c_int32), ("shard_count", C.c_int32),
                ("device", C.c_int32), ("flags", C.c_int32), ("stream", C.c_void_p),
                ("integrator", C.c_uint32), ("emitter_samples", C.c_int32), ("bsdf_samples", C.c_int32),
                ("n_devices", C.c_int32), ("devices", C.c_int32 * PHIP_MAX_DEVICES),
                ("sample_offset", C.c_int32), ("sample_total", C.c_int32),
                ("progress", PROGRESS_FN), ("progress_user", C.c_void_p),
                ("sobol_matrices", C.POINTER(C.c_uint32)), ("sobol_vdc", C.POINTER(C.c_uint64)), ("sobol_vdc_inv", C.POINTER(C.c_uint64)),
                ("sobol_dimensions", C.c_uint32), ("sobol_log_resolution", C.c_uint32), ("sobol_scramble", C.c_uint64),
                ("qmc_primes", C.POINTER(C.c_uint32)), ("qmc_permutations", C.POINTER(C.c_uint16)), ("qmc_dimensions", C.c_uint32), ("qmc_reserved", C.c_uint32)]


class phip_stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("closest_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("path_vertices", C.c_uint64), ("closest_node_visits", C.c_uint64), ("closest_triangle_tests", C.c_uint64),
                ("shadow_node_visits", C.c_uint64), ("shadow_triangle_tests", C.c_uint64),
                ("invalid_samples", C.c_uint64), ("iterations", C.c_uint32), ("vertex_traced", C.c_uint32),
                ("render_ms", C.c_double), ("trace_kernel_ms", C.c_double), ("shadow_kernel_ms", C.c_double),
                ("shade_kernel_ms", C.c_double), ("film_kernel_ms", C.c_double),
                ("algorithmic_bytes", C.c_double), ("trace_kernel_bytes", C.c_double),
                ("fused_kernel_ms", C.c_double), ("reduce_ms", C.c_double), ("fused", C.c_uint32), ("n_devices", C.c_uint32),
                ("d2h_ms", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if not n.startswith("reserved")}


class phip_ray(C.Structure):
    _fields_ = [("o", C.c_float * 3), ("mint", C.c_float), ("d", C.c_float * 3), ("maxt", C.c_float)]


class phip_hit(C.Structure):
    _fields_ = [("t", C.c_float), ("u", C.c_float), ("v", C.c_float), ("prim", C.c_uint32)]


class phip_accel_info(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("n_leaves", C.c_uint32), ("n_triangle_refs", C.c_uint32),
                ("max_depth", C.c_uint32), ("node_bytes", C.c_uint32), ("triangle_bytes", C.c_uint32),
                ("sah_cost", C.c_float), ("build_ms", C.c_float), ("fits_lds", C.c_uint32), ("fused_traversal", C.c_uint32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def default_render_params(**kw):
    """MonteCarloIntegrator defaults (src/librender/integrator.cpp:190-225)."""
    p = phip_render_params()
    p.spp = 4
    p.max_depth = -1
    p.rr_depth = 5
    p.strict_normals = 0
    p.hide_emitters = 0
    p.block_size = 32
    p.sampler = PHIP_SAMPLER_CTR
    p.seed = 0
    p.shard_index = 0
    p.shard_count = 1
    p.device = 0
    p.flags = 0
    p.stream = None
    p.integrator = PHIP_INTEGRATOR_PATH
    p.emitter_samples = 1
    p.bsdf_samples = 1
    p.n_devices = 0
    p.sample_offset = 0
    p.sample_total = 0
    devices = kw.pop("devices", None)
    if devices is not None:
        p.n_devices = len(devices)
        for i, d in enumerate(devices):
            p.devices[i] = d
    sobol = kw.pop("sobol", None)      # (matrices32, vdc, vdc_inv, log_resolution): numpy arrays out of the reference's sobol plugin (oracle/ref_ffi.sobol_tables)
    if sobol is not None:
        matrices, vdc, inv, m = sobol
        p.sampler = PHIP_SAMPLER_SOBOL
        p.sobol_matrices = matrices.ctypes.data_as(C.POINTER(C.c_uint32))
        p.sobol_vdc = vdc.ctypes.data_as(C.POINTER(C.c_uint64)); p.sobol_vdc_inv = inv.ctypes.data_as(C.POINTER(C.c_uint64))
        p.sobol_dimensions = len(matrices) // PHIP_SOBOL_MATRIX_SIZE; p.sobol_log_resolution = m; p.sobol_scramble = 0
        p._keep_sobol = sobol                  # keep the arrays alive as long as the struct
    rinv = kw.pop("qmc", None)         # (primes, permutations or None): numpy arrays out of the reference (oracle/ref_ffi.qmc_tables); the caller sets sampler=
    if rinv is not None:
        primes, perm = rinv
        p.qmc_primes = primes.ctypes.data_as(C.POINTER(C.c_uint32)); p.qmc_dimensions = len(primes)
        p.qmc_permutations = perm.ctypes.data_as(C.POINTER(C.c_uint16)) if perm is not None else None
        p._keep_qmc = rinv
    progress = kw.pop("progress", None)
    if progress is not None:
        p.progress = progress if isinstance(progress, PROGRESS_FN) else PROGRESS_FN(progress)
        p._keep_progress = p.progress          # keep the thunk alive as long as the struct
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p
