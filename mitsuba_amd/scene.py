"""Host-side scene assembly for the `path_hip` back end.

`SceneBuilder` plays the role of the Mitsuba plugin shim's flattening step
(mitsuba_amd/plugin/path_hip.cpp walks Scene::getShapes()/getEmitters()/getSensor() and fills
the same phip_scene_desc): triangle meshes with a material and an optional area emitter, a
perspective sensor and an hdrfilm with a reconstruction filter.  Names follow the reference's
XML vocabulary (shape / bsdf / emitter / sensor / film / rfilter).

Benchmark scenes (BASELINE.json configs; the reference ships no Cornell/Sponza assets, so they
are generated deterministically here):
    cornell_box()      -- classic measured Cornell box, diffuse + quad area light   (C1, C2)
    atrium()           -- "Sponza-class" procedural atrium, ~260k triangles         (C3, C5)
    glass_room()       -- tiled room with tessellated glass objects, ~150k tris     (C4)
"""
import ctypes as C
import math
import numpy as np

from . import _abi as A


def _f32(x):
    return np.ascontiguousarray(x, dtype=np.float32)


def look_at(origin, target, up):
    """Transform::lookAt (src/libcore/transform.cpp:191-214): camera looks down +z, x = left."""
    p = np.asarray(origin, np.float64)
    t = np.asarray(target, np.float64)
    u = np.asarray(up, np.float64)
    d = t - p
    d /= np.linalg.norm(d)
    left = np.cross(u, d)
    left /= np.linalg.norm(left)
    new_up = np.cross(d, left)
    m = np.eye(4)
    m[:3, 0] = left
    m[:3, 1] = new_up
    m[:3, 2] = d
    m[:3, 3] = p
    return m.astype(np.float32)


class SceneBuilder:
    def __init__(self):
        self.positions = []      # list of (n,3) float32
        self.normals = []        # list of (n,3) float32 or None
        self.indices = []        # list of (m,3) uint32 (local to the mesh)
        self.shapes = []         # dicts
        self.materials = []      # phip_material
        self.emitters = []       # dicts
        self.camera = None
        self.film = None
        self._keep = []

    # ---- bsdfs -------------------------------------------------------------------------
    def _add_material(self, m):
        self.materials.append(m)
        return len(self.materials) - 1

    def diffuse(self, reflectance=(0.5, 0.5, 0.5)):
        m = A.phip_material()
        m.type = A.PHIP_BSDF_DIFFUSE
        m.reflectance[:] = _rgb(reflectance)
        return self._add_material(m)

    def dielectric(self, int_ior=1.5046, ext_ior=1.000277, specular_reflectance=1.0, specular_transmittance=1.0):
        """defaults: intIOR bk7, extIOR air (src/bsdfs/dielectric.cpp:149-152, ior.h)"""
        m = A.phip_material()
        m.type = A.PHIP_BSDF_DIELECTRIC
        m.eta[0] = np.float32(np.float32(int_ior) / np.float32(ext_ior))
        m.reflectance[:] = _rgb(specular_reflectance)
        m.transmittance[:] = _rgb(specular_transmittance)
        return self._add_material(m)

    def roughconductor(self, eta, k, alpha=0.1, alpha_v=None, distribution="beckmann", sample_visible=True,
                       specular_reflectance=1.0, ext_eta=1.000277):
        """eta/k are linear-RGB (the host converts data/ior/*.spd, roughconductor.cpp:176-190)."""
        m = A.phip_material()
        m.type = A.PHIP_BSDF_ROUGHCONDUCTOR
        e = np.float32(ext_eta)
        m.eta[:] = [float(np.float32(x) / e) for x in _rgb(eta)]
        m.k[:] = [float(np.float32(x) / e) for x in _rgb(k)]
        m.alpha_u = alpha
        m.alpha_v = alpha if alpha_v is None else alpha_v
        m.distribution = {"beckmann": A.PHIP_MF_BECKMANN, "ggx": A.PHIP_MF_GGX}[distribution]
        m.sample_visible = 1 if sample_visible else 0
        m.reflectance[:] = _rgb(specular_reflectance)
        return self._add_material(m)

    def twosided(self, front, back=None):
        m = A.phip_material()
        m.type = A.PHIP_BSDF_TWOSIDED
        m.nested[0] = front
        m.nested[1] = front if back is None else back
        return self._add_material(m)

    # ---- shapes ------------------------------------------------------------------------
    def mesh(self, positions, triangles, material, normals=None, radiance=None, sampling_weight=1.0):
        positions = _f32(positions).reshape(-1, 3)
        triangles = np.ascontiguousarray(triangles, dtype=np.uint32).reshape(-1, 3)
        if normals is not None:
            normals = _f32(normals).reshape(-1, 3)
            assert normals.shape == positions.shape
        sid = len(self.shapes)
        emitter = -1
        if radiance is not None:
            emitter = len(self.emitters)
            self.emitters.append({"radiance": _rgb(radiance), "weight": sampling_weight, "shape": sid})
        self.positions.append(positions)
        self.normals.append(normals)
        self.indices.append(triangles)
        self.shapes.append({"material": material, "emitter": emitter})
        return sid

    def quad(self, p0, p1, p2, p3, material, facing=None, radiance=None):
        """Two triangles (0,1,2),(2,3,0) like Rectangle::createTriMesh (rectangle.cpp:170-203);
        if `facing` is given the winding is flipped so the face normal points that way."""
        P = _f32([p0, p1, p2, p3])
        n = np.cross(P[1] - P[0], P[2] - P[0])
        if facing is not None and np.dot(n, np.asarray(facing, np.float32)) < 0:
            P = P[::-1].copy()
        return self.mesh(P, [[0, 1, 2], [2, 3, 0]], material, radiance=radiance)

    # ---- sensor / film ------------------------------------------------------------------
    def perspective(self, origin, target, up, fov_x_deg, near=1e-2, far=1e4):
        cam = A.phip_camera()
        cam.to_world[:] = look_at(origin, target, up).reshape(-1).tolist()
        cam.xfov_deg = fov_x_deg
        cam.near_clip = near
        cam.far_clip = far
        self.camera = cam

    def hdrfilm(self, width, height, filter_table, crop=None):
        """filter_table = (radius, [32 floats]) from a ReconstructionFilter provider."""
        f = A.phip_film()
        f.width, f.height = width, height
        if crop is None:
            crop = (0, 0, width, height)
        f.crop_offset_x, f.crop_offset_y, f.crop_width, f.crop_height = crop
        f.filter_radius = filter_table[0]
        f.filter_table[:] = list(filter_table[1])
        self.film = f

    # ---- flatten -----------------------------------------------------------------------
    def desc(self):
        d = A.phip_scene_desc()
        d.abi_version = A.PHIP_ABI_VERSION
        pos = np.concatenate(self.positions).astype(np.float32) if self.positions else np.zeros((0, 3), np.float32)
        any_normals = any(n is not None for n in self.normals)
        nrm = None
        if any_normals:
            nrm = np.concatenate([n if n is not None else np.zeros_like(p) for n, p in zip(self.normals, self.positions)]).astype(np.float32)
        idx, shapes = [], (A.phip_shape * max(1, len(self.shapes)))()
        v0 = t0 = 0
        for i, (p, t, s) in enumerate(zip(self.positions, self.indices, self.shapes)):
            idx.append(t + np.uint32(v0))
            sh = shapes[i]
            sh.first_vertex, sh.n_vertices = v0, len(p)
            sh.first_triangle, sh.n_triangles = t0, len(t)
            sh.material, sh.emitter = s["material"], s["emitter"]
            sh.has_normals = 1 if self.normals[i] is not None else 0
            v0 += len(p)
            t0 += len(t)
        idx = np.ascontiguousarray(np.concatenate(idx), dtype=np.uint32) if idx else np.zeros((0, 3), np.uint32)
        pos = np.ascontiguousarray(pos)
        mats = (A.phip_material * max(1, len(self.materials)))(*self.materials)
        ems = (A.phip_emitter * max(1, len(self.emitters)))()
        for i, e in enumerate(self.emitters):
            ems[i].radiance[:] = e["radiance"]
            ems[i].sampling_weight = e["weight"]
            ems[i].shape = e["shape"]
        d.n_vertices = len(pos)
        d.positions = pos.ctypes.data_as(C.POINTER(C.c_float))
        d.normals = nrm.ctypes.data_as(C.POINTER(C.c_float)) if nrm is not None else None
        d.n_triangles = len(idx)
        d.indices = idx.ctypes.data_as(C.POINTER(C.c_uint32))
        d.n_shapes, d.shapes = len(self.shapes), shapes
        d.n_materials, d.materials = len(self.materials), mats
        d.n_emitters, d.emitters = len(self.emitters), ems
        d.camera, d.film = self.camera, self.film
        d._keep = (pos, nrm, idx, shapes, mats, ems)   # keep the buffers alive with the struct
        return d

    @property
    def n_triangles(self):
        return sum(len(t) for t in self.indices)


def _rgb(v):
    if np.isscalar(v):
        return [float(v)] * 3
    v = list(v)
    assert len(v) == 3
    return [float(x) for x in v]


# ==========================================================================================
#  C1 / C2: the Cornell box (classic measured data, units of mm)
# ==========================================================================================
def cornell_box(width, height, filter_table, light_scale=1.0, sb=None):
    sb = sb or SceneBuilder()
    white = sb.diffuse((0.725, 0.71, 0.68))
    red = sb.diffuse((0.63, 0.065, 0.05))
    green = sb.diffuse((0.14, 0.45, 0.091))
    light_bsdf = sb.diffuse((0.78, 0.78, 0.78))

    # room (normals face inward)
    sb.quad((552.8, 0, 0), (0, 0, 0), (0, 0, 559.2), (549.6, 0, 559.2), white, facing=(0, 1, 0))            # floor
    sb.quad((556.0, 548.8, 0), (556.0, 548.8, 559.2), (0, 548.8, 559.2), (0, 548.8, 0), white, facing=(0, -1, 0))  # ceiling
    sb.quad((549.6, 0, 559.2), (0, 0, 559.2), (0, 548.8, 559.2), (556.0, 548.8, 559.2), white, facing=(0, 0, -1))  # back
    sb.quad((0, 0, 559.2), (0, 0, 0), (0, 548.8, 0), (0, 548.8, 559.2), green, facing=(1, 0, 0))             # right (x=0)
    sb.quad((552.8, 0, 0), (549.6, 0, 559.2), (556.0, 548.8, 559.2), (556.0, 548.8, 0), red, facing=(-1, 0, 0))    # left
    # area light just below the ceiling
    sb.quad((343.0, 548.7, 227.0), (343.0, 548.7, 332.0), (213.0, 548.7, 332.0), (213.0, 548.7, 227.0), light_bsdf,
            facing=(0, -1, 0), radiance=tuple(light_scale * c for c in (17.0, 12.0, 4.0)))

    def box(top, h):
        """top: 4 corner (x,z) pairs of the top face, counter-clockwise seen from above"""
        top = [np.array([x, h, z], np.float32) for x, z in top]
        bot = [np.array([p[0], 0.0, p[2]], np.float32) for p in top]
        c = sum(top) / 4.0
        cx = np.array([c[0], h / 2.0, c[2]], np.float32)
        sb.quad(top[0], top[1], top[2], top[3], white, facing=(0, 1, 0))
        for i in range(4):
            j = (i + 1) % 4
            mid = (top[i] + top[j] + bot[i] + bot[j]) / 4.0
            sb.quad(top[i], bot[i], bot[j], top[j], white, facing=mid - cx)

    box([(130.0, 65.0), (82.0, 225.0), (240.0, 272.0), (290.0, 114.0)], 165.0)     # short box
    box([(423.0, 247.0), (265.0, 296.0), (314.0, 456.0), (472.0, 406.0)], 330.0)   # tall box

    sb.perspective(origin=(278.0, 273.0, -800.0), target=(278.0, 273.0, -799.0), up=(0, 1, 0),
                   fov_x_deg=39.3077, near=10.0, far=2800.0)
    sb.hdrfilm(width, height, filter_table)
    return sb
