"""Register budgets of the hot kernels, read from the compiler (hipcc -Rpass-analysis=kernel-resource-usage; cross-compiles, no GPU).

The persistent ray kernel's speed follows its resident waves (HISTORY.md 3.4: 4 / 5 / 6 waves per SIMD = 188.7 / 169.1 / 158.9 ms per C3
frame), so its VGPR count is a design property, not an accident of the compiler: 7 waves per SIMD (round 4) need <= 72 VGPRs and no scratch.  Its SGPR
count decides how many 256-thread blocks a CU admits (MI355X guide: 82..96 SGPRs -> 7 blocks, whatever the occupancy API answers); the
persistent grid is sized for WIDE_WAVES blocks per CU, so the count must admit that many or the surplus blocks run in a second round."""
import os
import re
import subprocess

import pytest

from mitsuba_amd import _ffi

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")


def resources(unit, extra=()):
    """{kernel name: {"vgprs": .., "sgprs": .., "scratch": .., "lds": ..}} of one translation unit (device code only, nothing is written)"""
    flags = [f for f in _ffi.HIPCC_FLAGS if f != "-shared"]
    cmd = [HIPCC] + flags + list(extra) + ["-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", os.path.join(_ffi.CSRC, unit), "-o", os.devnull]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        for key, pat in (("vgprs", r" VGPRs: (\d+)"), ("sgprs", r"TotalSGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    return out


def blocks_per_cu_by_sgprs(sgprs):
    """256-thread blocks a CU admits by its scalar registers: floor(800 / (ceil(sgpr / 16) * 16 + 16)), at most 8"""
    return min(8, 800 // (-(-sgprs // 16) * 16 + 16))


def test_ray_kernel_of_the_big_scenes_fits_seven_waves_per_simd():
    res = resources("phip.hip")
    k = next(v for name, v in res.items() if name.startswith("_Z8k_rays_w"))
    src = open(os.path.join(_ffi.CSRC, "k_wide_node.h")).read() + open(os.path.join(_ffi.CSRC, "k_wide.h")).read()      # (round 6: the tree's constants moved to k_wide_node.h)
    waves = int(re.search(r"#define WIDE_WAVES (\d+)", src).group(1))
    assert waves == 7
    assert k["scratch"] == 0, k
    assert k["vgprs"] <= 512 // waves // 8 * 8, k                  # 72: the allocation granule is 8 registers
    assert blocks_per_cu_by_sgprs(k["sgprs"]) >= waves, k           # the persistent grid is sized for `waves` blocks per CU
    stack = int(re.search(r"#define WIDE_STACK_LDS (\d+)", src).group(1)); cache = int(re.search(r"#define WIDE_NODE_CACHE_MAX (\d+)", src).group(1))
    deal = 4 * (64 * 8 + 64 * 8 + 256 * 2) if re.search(r"#define WIDE_DEAL 1", src) else 0      # WD_WAVE_BYTES per wave: slots, (u, v), work list
    assert waves * (stack * 8 * 256 + cache * 80 + deal + k["lds"]) <= 160 * 1024      # ... and their LDS (stack + node cache + triangle rounds + counters) fits the CU
    # the standalone ray cast of phip_trace on the same tree
    rc = next(v for name, v in res.items() if name.startswith("_Z11k_raycast_w"))
    assert rc["scratch"] == 0 and rc["vgprs"] <= 80, rc


def test_fused_kernel_keeps_four_waves_without_scratch():
    flags = next(u[1] for u in _ffi.UNITS if u[0] == "phip_mega.hip")
    res = resources("phip_mega.hip", flags)
    flat = [v for name, v in res.items() if name.startswith("_Z6k_megaILi0ELb0ELi2ELb0E")]          # diffuse, no strictNormals, packed flat table + record masks: C2's kernel
    assert flat and flat[0]["vgprs"] <= 128 and flat[0]["scratch"] == 0, flat
    n = 0
    for name, v in res.items():
        if name.startswith("_Z6k_mega"):
            n += 1
            qmc = re.match(r"_Z6k_megaILi\dELb[01]ELi\dELb1E", name) is not None            # the QMC builds
            packed = re.match(r"_Z6k_megaILi\dELb[01]ELi[23]E", name) is not None             # the packed leaf table (C2's class of scenes)
            allmat = name.startswith("_Z6k_megaILi3E")                                        # round 5: all three leaf BSDF models (packed table only)
            # diffuse builds: no scratch with the counter stream; the QMC build on the packed table parks 2 dwords (6 with strictNormals) since the interleaved 2D
            # radical inverse (halton 3001 -> 3226 Msamples/s for them; it was 36-64 B before the device drew Sobol' numbers through the byte tables only -- the row
            # loops' 16 reads in flight were what spilled); BVH4 walk / leaf table: up to 10 dwords.  All-material builds: the microfacet code
            # beside the traversal and the mailbox protocol parks 10-14 dwords (counter stream; 3-8 before the mailboxes, which are worth +26 % on the mixed box), ~27 with the samplers' tables live AND the mailboxes
            # (round 6: worth +24 % with sobol, +21 % with halton on the mixed box) -- at four waves per SIMD all the same (measured: the mixed
            # Cornell box 2360 Msamples/s on this kernel against 1770 on the one-kernel iterations at five waves)
            strict = re.match(r"_Z6k_megaILi\dELb1E", name) is not None
            limit = (112 if qmc else 64) if allmat else (((40 if strict else 32) if not packed else (24 if strict else 8)) if qmc else 0)
            assert v["vgprs"] <= 128 and v["scratch"] <= limit, (name, v)
            assert 4 * (v["lds"] + 12 * 1024) <= 160 * 1024, (name, v)      # four blocks per CU with the Cornell box's 11 KB of dynamic LDS (tables, records, flat table)
    assert n == 16                                                  # {diffuse, all materials} x strictNormals x {packed flat table of <= 32 / <= 64 records} x {counter stream, QMC} (round 6: the BVH4 walk and the per-lane leaf table in LDS are experiment builds -- scenes past 64 records are on the 8-wide tree)


def test_shading_kernels_of_the_metric_configurations_keep_their_waves():
    """k_shade of scenes without environment emitter / textures (FEAT bits 2 and 4: LDS-addressed tables) runs five waves per SIMD where more than one BSDF
    model is present (SHADE_WAVES_PLAIN; <= 96 VGPRs, its LDS of 28.5 KB admits five blocks per CU) and the lean diffuse instantiation fits 88.  Scratch is
    bounded: the allocator may park a few dwords outside the vertex's hot path, not more (HISTORY.md 3.4: k_shade forced to more waves with spills lost)."""
    res = {}
    for part in (0, 1):                                             # k_shade without / with strictNormals: two objects of the feature set
        res.update(resources("phip_shade.hip", next(u[1] for u in _ffi.UNITS if u[2] == "phip_shade0_%d.o" % part)))
    seen = 0
    for name, v in res.items():
        m = re.match(r"_Z7k_shadeILi(\d)ELb([01])ELi(\d+)E", name)
        if not m or int(m.group(3)) not in (4, 16):
            continue
        seen += 1
        mm = int(m.group(1))
        assert v["vgprs"] <= (88 if mm == 0 else 96), (name, v)
        assert v["scratch"] <= (64 if (mm in (1, 2) and m.group(2) == "0") else 96), (name, v)      # (1, 2, no strictNormals: the atrium's and the glass room's kernels)
        assert 5 * v["lds"] <= 160 * 1024, (name, v)
    assert seen == 16                                               # 4 material sets x strictNormals x {both tables in LDS, the emitter table only}


def test_film_splat_keeps_its_accumulators_in_registers():
    """k_film_splat<2, false> (gaussian / tent defaults, counter stream: the film pass of every configuration of the metric) holds 125 partial sums per thread:
    two waves per SIMD (256 VGPRs) and no scratch -- a branch around the accumulation or hoisted per-pixel invariants push them into scratch"""
    res = resources("phip.hip")
    k = next(v for name, v in res.items() if name.startswith("_Z12k_film_splatILi2ELb0E"))
    assert k["scratch"] == 0 and k["vgprs"] <= 256, k
    k1 = next(v for name, v in res.items() if name.startswith("_Z12k_film_splatILi1ELb0E"))
    assert k1["scratch"] == 0, k1


def test_vertex_and_rays_kernel_of_the_small_scenes_keeps_five_waves():
    """k_shade_trace (round 5: vertex + shadow ray + next ray per slot and launch, scenes on the packed leaf table that k_mega does not serve): 96 VGPRs = five
    waves per SIMD (measured +7 % over four on the mixed Cornell box), its static LDS -- the exchange buffer of the class deal, which the dealt traversals
    reuse -- leaves room for five blocks per CU beside ~6 KB of tables"""
    res = resources("phip_shade.hip", next(u[1] for u in _ffi.UNITS if u[2] == "phip_shade0_3.o"))
    ks = {n: v for n, v in res.items() if n.startswith("_Z13k_shade_trace")}
    assert len(ks) == 4, list(ks)
    for name, v in ks.items():
        assert v["vgprs"] <= 96 and v["scratch"] <= 64, (name, v)
        assert 5 * (v["lds"] + 8 * 1024) <= 160 * 1024, (name, v)


def scratch_instructions(unit, extra, prefix):
    """{mangled kernel name: number of scratch_load / scratch_store instructions in its ISA} for the kernels of `unit` whose name starts with `prefix` (hipcc -S, device only)"""
    import tempfile
    flags = [f for f in _ffi.HIPCC_FLAGS if f != "-shared"]
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        r = subprocess.run([HIPCC] + flags + list(extra) + ["--cuda-device-only", "-S", "-c", os.path.join(_ffi.CSRC, unit), "-o", out], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        counts, cur = {}, None
        for line in open(out):
            m = re.match(r"^(_Z\w+):", line)
            if m:
                cur = m.group(1) if m.group(1).startswith(prefix) else None
                if cur:
                    counts[cur] = 0
            elif cur and re.search(r"\bscratch_(load|store)", line):
                counts[cur] += 1
    return counts


def test_shading_kernels_of_the_metric_configurations_execute_no_scratch_instruction():
    """VERDICT r5 item 3 read the compiler's `ScratchSize` of 36-84 B per lane as spilling in the vertex kernel's hot path.  For the kernels of the metric's configurations it is
    not: the 36 bytes every k_shade reports are the frame the backend reserves when it parks SGPRs in VGPR lanes (v_writelane: 106 SGPRs in use) -- the ISA of the diffuse
    kernels and of the atrium's k_shade<MM_ROUGH, false, 16> (C3, C5) contains NO scratch instruction, that of the glass room's k_shade<MM_DIELECTRIC, false, 4> (C4) two
    (one dwordx3 parked outside the vertex's loop-free hot path).  This test pins the instruction counts, which is what the vector-memory path sees."""
    flags = next(u[1] for u in _ffi.UNITS if u[2] == "phip_shade0_0.o")
    counts = scratch_instructions("phip_shade.hip", flags, "_Z7k_shadeILi")
    seen = 0
    for name, n in counts.items():
        m = re.match(r"_Z7k_shadeILi(\d)ELb0ELi(\d+)E", name)
        if not m:
            continue
        mm, feat = int(m.group(1)), int(m.group(2))
        if mm == 0 or (mm == 1 and feat in (0, 16)):
            assert n == 0, (name, n)                      # diffuse scenes (C1, C2 under PHIP_FLAG_NO_FUSED), the atrium (C3, C5)
            seen += 1
        elif feat in (4, 16):
            assert n <= 4 if mm == 2 else n <= 20, (name, n)      # the glass room (C4): 2; all three models: a dozen and a half
    assert seen >= 5, counts


def test_fused_kernel_on_the_wide_tree_keeps_four_blocks_per_cu():
    """round 6: k_mega<.., FLAT 4 / 5, ..> (the tree in memory, one shared task stack per wave: k_wide_wave.h).  Four waves per SIMD (<= 128 VGPRs), the diffuse counter-stream
    builds without scratch, and a block's LDS -- static + task stacks + 48 cached nodes + slots / ray table / pair list of four waves + ~1.5 KB of tables -- leaves room for four
    blocks per CU: measured, three blocks (a 112-node cache) cost 14 % (profiles/r06_gpu_call_f_*)."""
    flags = next(u[1] for u in _ffi.UNITS if u[2] == "phip_megaw.o")
    res = resources("phip_mega.hip", flags)
    src = open(os.path.join(_ffi.CSRC, "k_wide_wave.h")).read()
    cap = int(re.search(r"#define WP_CAP (\d+)u", src).group(1)); pairs = int(re.search(r"#define WP_PAIRS (\d+)u", src).group(1))
    dyn = 4 * cap * 8 + 48 * 80 + 4 * (128 * 8 + 64 * 8 + 64 * 32 + pairs * 4) + 1536
    n = 0
    for name, v in res.items():
        m = re.match(r"_Z6k_megaILi(\d)ELb([01])ELi([45])ELb([01])E", name)
        if not m:
            continue
        n += 1
        assert v["vgprs"] <= 128, (name, v)
        if m.group(1) == "0" and m.group(4) == "0":
            assert v["scratch"] == 0, (name, v)
        blocks = 3 if m.group(4) == "1" else 4                      # (the QMC builds' camera-sample queue carries the sequence index: 4.4 KB more static LDS, three blocks)
        assert blocks * (v["lds"] + dyn) <= 160 * 1024, (name, v, dyn)
    assert n == 16              # {diffuse, all models} x strictNormals x {materials in LDS, in memory} x {counter stream, QMC}
