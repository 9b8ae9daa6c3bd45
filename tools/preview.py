"""Renders a scene with the CPU oracle at low resolution and writes a PNG (no GPU needed)."""
import sys, time, os, zlib, struct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mitsuba_amd import scene as S, _ffi, _abi as A
from oracle import oracle_ffi as O


def write_png(path, img):
    h, w, _ = img.shape
    raw = b''.join(b'\x00' + img[y].tobytes() for y in range(h))
    def chunk(t, d):
        c = struct.pack('>I', len(d)) + t + d
        return c + struct.pack('>I', zlib.crc32(t + d) & 0xffffffff)
    open(path, 'wb').write(b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, 2, 0, 0, 0)) + chunk(b'IDAT', zlib.compress(raw)) + chunk(b'IEND', b''))


if __name__ == "__main__":
    name = sys.argv[1]; w = int(sys.argv[2]); h = int(sys.argv[3]); spp = int(sys.argv[4]); md = int(sys.argv[5])
    ft = _ffi.gaussian_filter()
    t = time.time(); sb = getattr(S, name)(w, h, ft); d = sb.desc()
    print(name, sb.n_triangles, "tris", len(sb.shapes), "shapes", len(sb.materials), "materials", "%.2fs" % (time.time() - t))
    t = time.time(); osc = O.OracleScene(d); k = osc.kd_info()
    print("  kd build %.2fs nodes %d indices %d depth %d expTrav %.1f expPrims %.1f" % (time.time() - t, k.n_nodes, k.n_indices, k.max_depth, k.exp_traversal_steps, k.exp_prims_intersected))
    t = time.time(); film, _, st = osc.render(A.default_render_params(spp=spp, max_depth=md))
    print("  render %.2fs" % (time.time() - t), {k_: v for k_, v in st.as_dict().items() if v and 'ms' not in k_})
    rgb = O.develop(film); img = np.clip(rgb ** (1 / 2.2), 0, 1)
    write_png('/tmp/%s.png' % name, (img * 255).astype(np.uint8))
