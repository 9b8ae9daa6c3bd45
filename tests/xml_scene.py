"""Writes a SceneBuilder description as a Mitsuba 0.6 scene file (scene.xml + one OBJ mesh per shape): the input of the reference's
OWN loader (src/librender/scenehandler.cpp) and command-line front end (src/mitsuba/mitsuba.cpp), built by oracle/Makefile.ref as
oracle/_ref/mitsuba.  Property names are those the reference's plugins read (cf. oracle/ref_driver.cpp, which makes the same objects
through PluginManager::createObject by hand).  Numbers are written with 17 significant digits: every float32 survives strtod."""
import os

import numpy as np

from mitsuba_amd import _abi as A


def _f(x):
    return "%.17g" % float(x)


def _spec(v):
    return ", ".join(_f(c) for c in v)


def _bsdf(desc, i, indent):
    m = desc.materials[i]
    pad = " " * indent
    if m.type == A.PHIP_BSDF_DIFFUSE:
        return '%s<bsdf type="diffuse"><spectrum name="reflectance" value="%s"/></bsdf>\n' % (pad, _spec(m.reflectance))
    if m.type == A.PHIP_BSDF_DIELECTRIC:
        return ('%s<bsdf type="dielectric"><float name="intIOR" value="%s"/><float name="extIOR" value="1"/>'
                '<spectrum name="specularReflectance" value="%s"/><spectrum name="specularTransmittance" value="%s"/></bsdf>\n'
                % (pad, _f(m.eta[0]), _spec(m.reflectance), _spec(m.transmittance)))
    if m.type == A.PHIP_BSDF_ROUGHCONDUCTOR:
        return ('%s<bsdf type="roughconductor"><string name="material" value="none"/><spectrum name="eta" value="%s"/><spectrum name="k" value="%s"/>'
                '<float name="extEta" value="1"/><spectrum name="specularReflectance" value="%s"/><string name="distribution" value="%s"/>'
                '<float name="alphaU" value="%s"/><float name="alphaV" value="%s"/><boolean name="sampleVisible" value="%s"/></bsdf>\n'
                % (pad, _spec(m.eta), _spec(m.k), _spec(m.reflectance), "ggx" if m.distribution == A.PHIP_MF_GGX else "beckmann",
                   _f(m.alpha_u), _f(m.alpha_v), "true" if m.sample_visible else "false"))
    if m.type == A.PHIP_BSDF_TWOSIDED:
        inner = _bsdf(desc, m.nested[0], indent + 2)
        if m.nested[1] != m.nested[0]:
            inner += _bsdf(desc, m.nested[1], indent + 2)
        return '%s<bsdf type="twosided">\n%s%s</bsdf>\n' % (pad, inner, pad)
    raise ValueError("material type %d" % m.type)


def write_scene_xml(desc, outdir, integrator="path", integrator_props=None, sampler="independent", spp=16, sampler_props=None, stddev=0.5, mesh_format="obj"):
    """-> path of scene.xml.  Bitmap textures / environment maps are not written (the pin scenes that use them go through ref_driver)."""
    os.makedirs(outdir, exist_ok=True)
    if desc.n_textures or any(desc.emitters[i].type == A.PHIP_EMITTER_ENVMAP for i in range(desc.n_emitters)):
        raise ValueError("textures / envmaps are not written by this helper")
    P = np.ctypeslib.as_array(desc.positions, shape=(desc.n_vertices, 3))
    T = np.ctypeslib.as_array(desc.indices, shape=(desc.n_triangles, 3))
    N = np.ctypeslib.as_array(desc.normals, shape=(desc.n_vertices, 3)) if desc.normals else None
    UV = np.ctypeslib.as_array(desc.texcoords, shape=(desc.n_vertices, 2)) if desc.texcoords else None
    x = ['<?xml version="1.0" encoding="utf-8"?>\n<!-- written by tests/xml_scene.py -->\n<scene version="0.6.0">\n']
    props = dict(integrator_props or {})
    x.append('  <integrator type="%s">\n' % integrator)
    for k, v in props.items():
        kind = "boolean" if isinstance(v, bool) else ("integer" if isinstance(v, int) else ("float" if isinstance(v, float) else "string"))
        x.append('    <%s name="%s" value="%s"/>\n' % (kind, k, str(v).lower() if isinstance(v, bool) else v))
    x.append('  </integrator>\n')
    c, f = desc.camera, desc.film
    x.append('  <sensor type="perspective">\n    <transform name="toWorld"><matrix value="%s"/></transform>\n' % " ".join(_f(v) for v in c.to_world))
    x.append('    <float name="fov" value="%s"/><string name="fovAxis" value="x"/><float name="nearClip" value="%s"/><float name="farClip" value="%s"/>\n'
             % (_f(c.xfov_deg), _f(c.near_clip), _f(c.far_clip)))
    x.append('    <sampler type="%s"><integer name="sampleCount" value="%d"/>' % (sampler, spp))
    for k, v in (sampler_props or {}).items():
        kind = "boolean" if isinstance(v, bool) else ("integer" if isinstance(v, int) else "string")
        x.append('<%s name="%s" value="%s"/>' % (kind, k, str(v).lower() if isinstance(v, bool) else v))
    x.append('</sampler>\n')
    x.append('    <film type="hdrfilm"><integer name="width" value="%d"/><integer name="height" value="%d"/>' % (f.width, f.height))
    x.append('<integer name="cropOffsetX" value="%d"/><integer name="cropOffsetY" value="%d"/><integer name="cropWidth" value="%d"/><integer name="cropHeight" value="%d"/>'
             % (f.crop_offset_x, f.crop_offset_y, f.crop_width, f.crop_height))
    x.append('<string name="pixelFormat" value="rgb"/><string name="fileFormat" value="pfm"/><string name="componentFormat" value="float32"/><boolean name="banner" value="false"/>')
    x.append('<rfilter type="gaussian"><float name="stddev" value="%s"/></rfilter></film>\n  </sensor>\n' % _f(stddev))
    for i in range(desc.n_emitters):
        e = desc.emitters[i]
        if e.type == A.PHIP_EMITTER_CONSTANT:
            x.append('  <emitter type="constant"><spectrum name="radiance" value="%s"/><float name="samplingWeight" value="%s"/></emitter>\n' % (_spec(e.radiance), _f(e.sampling_weight)))
    for si in range(desc.n_shapes):
        s = desc.shapes[si]
        if mesh_format == "serialized":
            # Mitsuba's own mesh format, written by the reference's TriMesh::serialize and read back by its `serialized` loader plugin
            from oracle import ref_ffi
            name = "shape%d.serialized" % si
            ref_ffi.write_serialized(desc, si, os.path.join(outdir, name))
            x.append('  <shape type="serialized"><string name="filename" value="%s"/>' % name)
            if not (s.has_normals and N is not None):
                x.append('<boolean name="faceNormals" value="true"/>')
            x.append('\n' + _bsdf(desc, s.material, 4))
            if s.emitter >= 0:
                e = desc.emitters[s.emitter]
                x.append('    <emitter type="area"><spectrum name="radiance" value="%s"/><float name="samplingWeight" value="%s"/></emitter>\n' % (_spec(e.radiance), _f(e.sampling_weight)))
            x.append('  </shape>\n')
            continue
        name = "shape%d.obj" % si
        with open(os.path.join(outdir, name), "w") as o:
            v0 = s.first_vertex
            for v in range(s.n_vertices):
                o.write("v %s %s %s\n" % tuple(_f(c) for c in P[v0 + v]))
            hn, ht = bool(s.has_normals and N is not None), bool(s.has_texcoords and UV is not None)
            if ht:
                for v in range(s.n_vertices):
                    o.write("vt %s %s\n" % tuple(_f(c) for c in UV[v0 + v]))
            if hn:
                for v in range(s.n_vertices):
                    o.write("vn %s %s %s\n" % tuple(_f(c) for c in N[v0 + v]))
            for t in range(s.n_triangles):
                idx = [int(k) - v0 + 1 for k in T[s.first_triangle + t]]
                o.write("f " + " ".join(("%d/%s/%s" % (k, k if ht else "", k if hn else "")) if (hn or ht) else "%d" % k for k in idx) + "\n")
        x.append('  <shape type="obj"><string name="filename" value="%s"/>' % name)
        if not (s.has_normals and N is not None):
            x.append('<boolean name="faceNormals" value="true"/>')
        x.append('\n' + _bsdf(desc, s.material, 4))
        if s.emitter >= 0:
            e = desc.emitters[s.emitter]
            x.append('    <emitter type="area"><spectrum name="radiance" value="%s"/><float name="samplingWeight" value="%s"/></emitter>\n' % (_spec(e.radiance), _f(e.sampling_weight)))
        x.append('  </shape>\n')
    x.append('</scene>\n')
    path = os.path.join(outdir, "scene.xml")
    open(path, "w").write("".join(x))
    return path


def read_pfm(path):
    """(H, W, 3) float32, top row first"""
    with open(path, "rb") as f:
        kind = f.readline().strip()
        w, h = (int(v) for v in f.readline().split())
        scale = float(f.readline())
        n = 3 if kind == b"PF" else 1
        a = np.frombuffer(f.read(), dtype="<f4" if scale < 0 else ">f4").reshape(h, w, n)
    return a[::-1].astype(np.float32)
