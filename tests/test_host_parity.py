"""CPU-only parity of the PRODUCT's shading arithmetic: the __host__ __device__ functions of
mitsuba_amd/csrc/dv_*.h compiled for the host (phip_debug_host_*) against the oracle, bit for bit.
(The same source compiled for gfx950 is compared on the GPU in test_gpu_parity.py / test_fmath.py.)"""
import ctypes as C

import numpy as np
import pytest

from mitsuba_amd import _abi as A, scene as S


def fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def zoo(gauss):
    sb = S.cornell_box(16, 16, gauss)
    cu = dict(eta=S.CU_ETA, k=S.CU_K)
    m = {"diffuse": 0,
         "rc beckmann": sb.roughconductor(alpha=0.3, **cu), "rc ggx aniso": sb.roughconductor(alpha=0.1, alpha_v=0.3, distribution="ggx", **cu),
         "rc beckmann aniso": sb.roughconductor(alpha=0.08, alpha_v=0.25, **cu), "rc non-visible": sb.roughconductor(alpha=0.05, sample_visible=False, **cu),
         "rc ggx non-visible": sb.roughconductor(alpha=0.2, sample_visible=False, distribution="ggx", **cu),
         "rc tiny alpha": sb.roughconductor(alpha=1e-6, **cu),
         "dielectric glass": sb.dielectric(1.5, 1.0), "dielectric water": sb.dielectric(1.333, 1.000277), "dielectric inverted": sb.dielectric(1.0, 1.5)}
    m["twosided diffuse"] = sb.twosided(0)
    m["twosided rc/diffuse"] = sb.twosided(m["rc beckmann"], 1)
    return sb, m


def test_bsdf_sample_eval_pdf_bitwise(oracle, phip, gauss):
    sb, mats = zoo(gauss)
    d = sb.desc(); osc = oracle.OracleScene(d); OL = oracle.lib()
    rng = np.random.default_rng(11); n = 100000
    wi = rng.normal(size=(n, 3)).astype(np.float32); wi /= np.linalg.norm(wi, axis=1, keepdims=True)
    wi[:100] = [0, 0, 1]; wi[100:200] = [0, 0, -1]; wi[200:300, 2] = 0; wi[200:300] /= np.linalg.norm(wi[200:300], axis=1, keepdims=True) + 1e-30
    smp = np.minimum(rng.random((n, 2)).astype(np.float32), np.float32(1) - np.float32(2 ** -24))   # [0, 1) like Random::nextFloat (a double close to 1 rounds to 1.0f); smp[:50] = 0; smp[50:100] = np.float32(1) - np.float32(2 ** -24)
    wo_r = rng.normal(size=(n, 3)).astype(np.float32); wo_r /= np.linalg.norm(wo_r, axis=1, keepdims=True)
    u8 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8))
    for name, mid in mats.items():
        a = [np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.float32), np.zeros(n, np.uint8)]
        b = [np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.float32), np.zeros(n, np.uint8)]
        assert phip.phip_debug_host_bsdf_sample(d.materials, d.n_materials, mid, n, fp(wi), fp(smp), fp(a[0]), fp(a[1]), fp(a[2]), u8(a[3])) == 0
        OL.oracle_bsdf_sample(osc.h, mid, n, fp(wi), fp(smp), fp(b[0]), fp(b[1]), fp(b[2]), u8(b[3]))
        for x, y in zip(a[:3], b[:3]):
            assert (x.view(np.uint32) == y.view(np.uint32)).all(), name
        nz = a[2] > 0
        assert (a[3][nz] == b[3][nz]).all(), name
        v1 = np.zeros((n, 3), np.float32); p1 = np.zeros(n, np.float32); v2 = np.zeros((n, 3), np.float32); p2 = np.zeros(n, np.float32)
        phip.phip_debug_host_bsdf_eval_pdf(d.materials, d.n_materials, mid, n, fp(wi), fp(wo_r), fp(v1), fp(p1))
        OL.oracle_bsdf_eval_pdf(osc.h, mid, n, fp(wi), fp(wo_r), fp(v2), fp(p2))
        assert (v1.view(np.uint32) == v2.view(np.uint32)).all() and (p1.view(np.uint32) == p2.view(np.uint32)).all(), name


def test_ctr_stream_bitwise_and_well_distributed(oracle, phip):
    """the parity stream: pcg4d(pixel, sample, block, seed) -> 4 floats in [0,1) with a 23-bit mantissa"""
    rng = np.random.default_rng(2)
    keys = rng.integers(0, 2 ** 32, (2000, 4), dtype=np.uint64).astype(np.uint32)
    keys[:64] = [[p, s, b, 0] for p in range(4) for s in range(4) for b in range(4)]
    out_o = np.zeros((len(keys), 4), np.float32); out_p = np.zeros_like(out_o)
    for i, (p, s, b, sd) in enumerate(keys):
        oracle.lib().oracle_ctr_block(int(p), int(s), int(b), int(sd), fp(out_o[i]))
        phip.phip_debug_host_ctr_block(int(p), int(s), int(b), int(sd), fp(out_p[i]))
    assert (out_o.view(np.uint32) == out_p.view(np.uint32)).all()
    assert (out_o >= 0).all() and (out_o < 1).all()
    assert (np.round(out_o.astype(np.float64) * 2 ** 23) == out_o.astype(np.float64) * 2 ** 23).all()
    # python restatement of pcg4d (Jarzynski & Olano 2020) as an independent third implementation
    def pcg4d(v):
        v = [(x * 1664525 + 1013904223) & 0xFFFFFFFF for x in v]
        m = 0xFFFFFFFF
        v[0] = (v[0] + v[1] * v[3]) & m; v[1] = (v[1] + v[2] * v[0]) & m; v[2] = (v[2] + v[0] * v[1]) & m; v[3] = (v[3] + v[1] * v[2]) & m
        v = [x ^ (x >> 16) for x in v]
        v[0] = (v[0] + v[1] * v[3]) & m; v[1] = (v[1] + v[2] * v[0]) & m; v[2] = (v[2] + v[0] * v[1]) & m; v[3] = (v[3] + v[1] * v[2]) & m
        return v
    for i in range(0, 200, 7):
        w = pcg4d([int(x) for x in keys[i]])
        f = ((np.array(w, np.uint32) >> 9) | np.uint32(0x3f800000)).view(np.float32) - np.float32(1)
        assert (f == out_o[i]).all()
    # neighbouring pixels / samples / dimensions are uncorrelated
    grid = np.zeros((64, 64, 4), np.float32)
    for p in range(64):
        for s in range(64):
            oracle.lib().oracle_ctr_block(p, s, 1, 0, fp(grid[p, s]))
    x = grid[..., 0].astype(np.float64)
    assert abs(x.mean() - 0.5) < 0.02 and abs(x.var() - 1 / 12) < 0.01
    for a, b in [(x[1:], x[:-1]), (x[:, 1:], x[:, :-1]), (grid[..., 0], grid[..., 1]), (grid[..., 2], grid[..., 3])]:
        assert abs(np.corrcoef(np.ravel(a), np.ravel(b))[0, 1]) < 0.05


def test_bvh_build_covers_every_triangle_and_scene_box_matches_kdtree(oracle, phip, gauss):
    """host BVH build (no GPU): every non-degenerate triangle referenced exactly once (scenes of 4096 triangles or more are built with
    spatial splits, bvh.h: at most 1.5 references per triangle); the enlarged scene box equals the kd-tree root box of the oracle
    (gkdtree.h:1213-1220 arithmetic)"""
    for sb in [S.cornell_box(16, 16, gauss), S.atrium(16, 16, gauss, detail=0.25), S.glass_room(16, 16, gauss, detail=0.3)]:
        d = sb.desc()
        info = A.phip_accel_info(); box = np.zeros(6, np.float32)
        assert phip.phip_debug_host_build_bvh(d.positions, d.n_vertices, d.indices, d.n_triangles, C.byref(info), fp(box)) == 0
        if d.n_triangles < 4096:
            assert info.n_triangle_refs == d.n_triangles        # no duplication, no degenerate input
        else:
            assert d.n_triangles <= info.n_triangle_refs <= 1.5 * d.n_triangles + 64
        assert info.n_leaves >= d.n_triangles / 8 and info.max_depth < 40
        k = oracle.OracleScene(d).kd_info()
        assert (box[:3].view(np.uint32) == np.array(list(k.aabb_min), np.float32).view(np.uint32)).all()
        assert (box[3:].view(np.uint32) == np.array(list(k.aabb_max), np.float32).view(np.uint32)).all()


def test_degenerate_and_empty_geometry_build(phip):
    p = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0], [0, 1, 0]], np.float32)       # first triangle is collinear
    idx = np.array([[0, 1, 2], [0, 1, 3]], np.uint32)
    info = A.phip_accel_info()
    assert phip.phip_debug_host_build_bvh(fp(p), 4, idx.ctypes.data_as(C.POINTER(C.c_uint32)), 2, C.byref(info), None) == 0
    assert info.n_triangle_refs == 1
    assert phip.phip_debug_host_build_bvh(fp(p), 4, idx.ctypes.data_as(C.POINTER(C.c_uint32)), 0, C.byref(info), None) == 0
    bad = np.array([[0, 1, 9]], np.uint32)
    assert phip.phip_debug_host_build_bvh(fp(p), 4, bad.ctypes.data_as(C.POINTER(C.c_uint32)), 1, C.byref(info), None) == A.PHIP_ERR_INVALID


def test_host_mip_eval_matches_the_oracle(oracle, phip):
    """MIPMap::eval as the kernels compile it (dv_scene.h, executed on the host) against the oracle's restatement -- itself
    pinned to the reference's TMIPMap -- on random pyramids, lookup parameters and footprints, including degenerate and
    non-finite differentials (NaN results must agree as well)"""
    import ctypes as C
    from mitsuba_amd import scene as S
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    rng = np.random.default_rng(1)
    L, Lo = phip, oracle.lib()
    for trial in range(60):
        w, h = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        keep = [np.ascontiguousarray(l, np.float32) for l in S.mip_pyramid(rng.uniform(0, 2, (h, w, 3)).astype(np.float32))]
        t = A.phip_texture(); t.width, t.height, t.n_levels = w, h, len(keep)
        for i, l in enumerate(keep):
            t.levels[i] = fp(l)
        t.wrap_u, t.wrap_v = int(rng.integers(0, 5)), int(rng.integers(0, 5))
        t.filter_type = int(rng.integers(0, 4)); t.max_anisotropy = float(rng.uniform(1, 20))
        n = 3000
        uv = rng.uniform(-2, 3, (n, 2)).astype(np.float32)
        sc = 10.0 ** rng.uniform(-6, 1, (n, 1))
        d0 = (rng.normal(size=(n, 2)) * sc).astype(np.float32)
        d1 = (rng.normal(size=(n, 2)) * sc * 10.0 ** rng.uniform(-3, 3, (n, 1))).astype(np.float32)
        d0[:50] = 0; d1[50:100] = 0; d1[100:150] = d0[100:150]; d1[150:200] = -d0[150:200] * 3
        d0[200:220] = np.nan; d1[220:240] = np.inf; d0[240:250] = 1e30; uv[250:260] = np.nan
        a = np.zeros((n, 3), np.float32); b = np.zeros((n, 3), np.float32)
        assert L.phip_debug_host_mip_eval(C.byref(t), n, fp(uv), fp(d0), fp(d1), fp(a)) == 0
        assert Lo.oracle_mip_eval(C.byref(t), n, fp(uv), fp(d0), fp(d1), fp(b)) == 0
        ok = (a.view(np.uint32) == b.view(np.uint32)).all(-1) | (np.isnan(a).all(-1) & np.isnan(b).all(-1))
        assert ok.all(), (trial, int((~ok).sum()))


def test_ld_sampler_points(oracle, phip):
    """PHIP_SAMPLER_LD: (1) the device code (compiled for the host) and the oracle produce the same points, bit for bit; (2) the n points
    of a (pixel, dimension) are a scrambled (0,2)-net -- every elementary interval of area 1/n holds exactly one of them -- and every
    sample index is used exactly once (the keyed order is a permutation); (3) different pixels and dimensions are scrambled
    differently; (4) the 1D requests are stratified too.  (ldsampler.cpp:151-186, core/qmc.h)"""
    O = oracle.lib()
    for n in (1, 2, 16, 64, 256):
        mask = n - 1
        for pixel, dim, seed in ((0, 0, 0), (12345, 2, 0), (7, 4, 3), (99999, 6, 1), (5, 1, 0), (5, 7, 9)):
            a = np.zeros((n, 2), np.float32); b = np.zeros((n, 2), np.float32)
            for k in range(n):
                O.oracle_ld_point(pixel, k, dim, seed, mask, fp(a[k]))
                phip.phip_debug_host_ld_point(pixel, k, dim, seed, mask, fp(b[k]))
            assert (a.view(np.uint32) == b.view(np.uint32)).all()
            assert (a[:, 0] >= 0).all() and (a[:, 0] < 1).all() and (a[:, 1] >= 0).all() and (a[:, 1] <= 1).all()
            x = np.minimum(a[:, 0].astype(np.float64), 1 - 1e-9); y = np.minimum(a[:, 1].astype(np.float64), 1 - 1e-9)
            m = n.bit_length() - 1
            for i in range(m + 1):                                  # elementary intervals 2^-i x 2^-(m-i)
                cells = np.floor(x * (1 << i)).astype(int) * (1 << (m - i)) + np.floor(y * (1 << (m - i))).astype(int)
                assert len(set(cells.tolist())) == n, (n, pixel, dim, i)
    pts = np.zeros((3, 16, 2), np.float32)
    for j, (pixel, dim) in enumerate(((1, 0), (2, 0), (1, 2))):
        for k in range(16):
            O.oracle_ld_point(pixel, k, dim, 0, 15, fp(pts[j, k]))
    assert not np.array_equal(pts[0], pts[1]) and not np.array_equal(pts[0], pts[2])


def test_cdf_sample_small_tables_equal_the_search(phip):
    """cdfSample (dv_scene.h) counts the elements below the value for tables of one to three entries instead of searching (the kernels'
    NEE samples do two such look-ups, each a chain of dependent reads): the index must be DiscreteDistribution::sample's
    (pmf.h:124-136: lower_bound - 1, clamped, zero-probability entries skipped) for every table size, ties and empty entries included."""
    L = phip
    rng = np.random.default_rng(11)

    def reference(cdf, n, v):
        lo = int(np.searchsorted(cdf[:n + 1], v, side="left"))       # std::lower_bound
        idx = min(max(lo - 1, 0), n - 1)
        while idx < n - 1 and np.float32(cdf[idx + 1]) - np.float32(cdf[idx]) == 0:
            idx += 1
        return idx

    for n in range(1, 8):
        for trial in range(40):
            pmf = rng.uniform(0, 1, n).astype(np.float32)
            pmf[rng.uniform(0, 1, n) < 0.3] = 0                         # entries of zero probability
            if pmf.sum() == 0:
                pmf[rng.integers(n)] = 1
            cdf = np.zeros(n + 1 + 3, np.float32)
            acc = np.float32(0)
            for i in range(n):
                acc = np.float32(acc + pmf[i]); cdf[i + 1] = acc
            cdf[:n + 1] /= cdf[n]; cdf[n] = 1.0
            cdf[n + 1:] = rng.uniform(-5, 5, 3)                          # whatever follows the table in memory must not matter
            vals = np.concatenate([rng.uniform(0, 1, 64).astype(np.float32), cdf[:n + 1], np.nextafter(cdf[:n + 1], 2).astype(np.float32),
                                   np.nextafter(cdf[:n + 1], -2).astype(np.float32), np.float32([0, 1])]).astype(np.float32)
            out = np.zeros(len(vals), np.uint32)
            L.phip_debug_host_cdf_sample(cdf.ctypes.data_as(C.POINTER(C.c_float)), n, vals.ctypes.data_as(C.POINTER(C.c_float)), len(vals),
                                         out.ctypes.data_as(C.POINTER(C.c_uint32)))
            want = np.array([reference(cdf, n, v) for v in vals], np.uint32)
            assert (out == want).all(), (n, cdf[:n + 1], vals[out != want], out[out != want], want[out != want])


def test_sobol_byte_tables_equal_the_row_loops(phip):
    """PHIP_SAMPLER_SOBOL: the device draws its numbers through 256-entry XOR tables per byte of the index (dv_math.h: SobolTab::matBt, built by the host from the
    plugin's direction numbers).  The same functions compiled for the host, with and without the tables, on the same random requests: every index of the pixel
    enumeration (sobol::look_up) and every number bit for bit; and the number against a direct restatement of sobol::sampleSingle (sobolseq.h:42-58) in numpy."""
    from conftest import sobol_tables
    rng = np.random.default_rng(11)
    u32p, u64p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    for (w, h), scramble in (((1024, 1024), 0), ((1920, 1080), 0x9E3779B9), ((2, 2), 0), ((4096, 2160), 123456789)):
        mat, vdc, vdc_inv, m = sobol_tables(w, h, dimensions=128)
        n = 20000
        sample = rng.integers(0, 1 << 12, n).astype(np.uint32); sample[:64] = np.arange(64)
        sample[64:96] = (1 << np.arange(32, dtype=np.uint64)).astype(np.uint32)                    # every byte of the frame number
        px = rng.integers(0, w, n).astype(np.uint32); py = rng.integers(0, h, n).astype(np.uint32)
        dim = rng.integers(0, 120, n).astype(np.uint32)
        res = []
        for bt in (0, 1):
            idx = np.zeros(n, np.uint64); val = np.zeros(n, np.float32); quad = np.zeros((n, 4), np.float32)
            rc = phip.phip_debug_host_sobol(mat.ctypes.data_as(u32p), len(mat) // 52, vdc.ctypes.data_as(u64p), vdc_inv.ctypes.data_as(u64p), m, scramble, bt, n,
                                            sample.ctypes.data_as(u32p), px.ctypes.data_as(u32p), py.ctypes.data_as(u32p), dim.ctypes.data_as(u32p),
                                            idx.ctypes.data_as(u64p), fp(val), fp(quad))
            assert rc == 0
            res.append((idx, val, quad))
        assert (res[0][0] == res[1][0]).all()
        assert (res[0][1].view(np.uint32) == res[1][1].view(np.uint32)).all()
        assert (res[0][2].view(np.uint32) == res[1][2].view(np.uint32)).all()
        assert (res[1][2][:, 0].view(np.uint32) == res[1][1].view(np.uint32)).all()                # the 2 x 2 form starts at the same dimension
        # sampleSingle restated: XOR of the rows of the set index bits, scaled by 2^-32, clamped below 1
        idx = res[0][0]
        acc = np.full(n, scramble, np.uint32)
        for bit in range(64):                                                                     # (above bit 51 sampleSingle reads on into the next dimension's rows)
            on = ((idx >> np.uint64(bit)) & np.uint64(1)).astype(bool)
            acc[on] ^= mat[dim[on].astype(np.int64) * 52 + bit]
        expect = np.minimum((acc.astype(np.float32) * np.float32(1.0 / 4294967296.0)), np.float32(0.99999994))
        assert (expect.view(np.uint32) == res[1][1].view(np.uint32)).all()
        if m > 1:
            assert len(np.unique(idx)) > n // 2 and (idx >> np.uint64(2 * m) == sample).all()      # look_up keeps the frame number in the bits above the pixel


def test_radical_inverse_tables_equal_the_digit_loops(phip):
    """PHIP_SAMPLER_HALTON / _HAMMERSLEY (round 5): the device draws its radical inverses through multi-digit tables (dv_math.h: RinvTab::chunk, built by the host:
    the permuted value of a chunk of k digits is one look-up, the division by base^k a multiply-high with a correction).  The same functions compiled for the
    host, with and without the tables, on random indices of every dimension -- and the digit loop against scrambledRadicalInverse (qmc.cpp:99-112) in numpy."""
    from conftest import qmc_tables
    rng = np.random.default_rng(23)
    u32p, u64p, u16p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint16)
    for scramble in (-1, 7, 0):
        primes, perm = qmc_tables(scramble, dimensions=64)
        primes = np.ascontiguousarray(primes, np.uint32)
        n = 60000
        index = rng.integers(0, 1 << 32, n, dtype=np.uint64)
        index[:4000] = np.arange(4000)                                                             # the first points, the empty index
        index[4000:8000] = rng.integers(0, 1 << 23, 4000, dtype=np.uint64)                          # C2's range (31104 x 256)
        for j, b in enumerate((2, 3, 5, 7, 11, 1024, 729, 625, 343, 121)):                          # multiples and neighbours of the chunk sizes
            index[8000 + 40 * j:8000 + 40 * j + 40] = (np.arange(1, 41, dtype=np.uint64) * np.uint64(b ** (1 + j % 3))) + np.uint64(j % 2) - np.uint64(j % 3 == 2)
        index[9000:9064] = (np.uint64(1) << np.arange(64, dtype=np.uint64))                         # ... incl. indices above 2^32 (the digit loops serve those)
        index[9064:9100] = np.uint64(0xFFFFFFFF) - np.arange(36, dtype=np.uint64)
        dim = rng.integers(0, len(primes), n).astype(np.uint32); dim[:4000] = np.arange(4000) % len(primes)
        res = []
        for tables in (0, 1):
            out = np.zeros(n, np.float32)
            rc = phip.phip_debug_host_rinv(primes.ctypes.data_as(u32p), perm.ctypes.data_as(u16p) if perm is not None else None, len(primes), tables, n,
                                           index.ctypes.data_as(u64p), dim.ctypes.data_as(u32p), fp(out))
            assert rc == 0
            res.append(out)
        assert (res[0].view(np.uint32) == res[1].view(np.uint32)).all(), np.argwhere(res[0] != res[1])[:10]
        assert (res[0] >= 0).all() and (res[0] < 1).all()
        # the interleaved 2D form (dimensions d, d + 1 in one pass: what a path vertex's 2D requests call), halton and hammersley numbering
        for ham in (0, 1):
            d2 = np.clip(dim, 1 + ham, len(primes) - 2).astype(np.uint32)
            out2 = np.zeros((n, 2), np.float32)
            assert phip.phip_debug_host_rinv2(primes.ctypes.data_as(u32p), perm.ctypes.data_as(u16p) if perm is not None else None, len(primes), ham, n,
                                              index.ctypes.data_as(u64p), d2.ctypes.data_as(u32p), fp(out2)) == 0
            for col in (0, 1):
                single = np.zeros(n, np.float32); base = (d2 + col - ham).astype(np.uint32)
                assert phip.phip_debug_host_rinv(primes.ctypes.data_as(u32p), perm.ctypes.data_as(u16p) if perm is not None else None, len(primes), 0, n,
                                                 index.ctypes.data_as(u64p), base.ctypes.data_as(u32p), fp(single)) == 0
                assert (single.view(np.uint32) == out2[:, col].view(np.uint32)).all(), (ham, col)
        # the definition (qmc.cpp:99-112, float accumulation) agrees to rounding: the Fast forms sum integer digits and scale once
        off = np.concatenate(([0], np.cumsum(primes.astype(np.int64))[:-1])).astype(np.int64)
        sel = np.arange(0, 3000)
        for i in sel[::7]:
            b = int(primes[dim[i]]); radical = 1.0 / b; inv = 0.0; digit = radical; t = int(index[i])
            P = (lambda d: int(perm[off[dim[i]] + d])) if perm is not None else (lambda d: d)
            while t:
                inv += digit * P(t % b); digit *= radical; t //= b
            if perm is not None:
                inv += digit * P(0) / (1 - radical)
            assert abs(min(inv, 0.99999994) - float(res[0][i])) <= 2e-6, (i, b, index[i], inv, res[0][i])
