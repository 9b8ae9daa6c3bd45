#!/bin/bash
# round 5, GPU call a (made on commit 973d15f: the material heads it measures were REMOVED afterwards, PHIP_NO_HEADS no longer exists): (1) material heads in the shading records (k_shade: two dependent round trips instead of three) against PHIP_NO_HEADS=1, and the
# k_rays_w variants built by tools/build_variant.sh -- `cull` (-DWIDE_CULL=1: popped node groups skip a child whose entry lies behind the hit),
# `b768c292w6` / `b1024c585w4` (blocks of 768 / 1024 lanes sharing an LDS cache of 292 / 585 top-of-tree nodes, read with ds_read_b128) -- on C3 / C4;
# (2) the small scenes: C2, the 42- / 62-record boxes (two-word record masks; PHIP_NO_FLAT3=1 = the per-lane leaf table they used before), the mixed box;
# (3) parity of what changed.   -> gpurun_out/r5a/
mkdir -p gpurun_out/r5a
o=gpurun_out/r5a
WORKLOADS="atrium 64;glass 128" AB_ENV="noheads PHIP_NO_HEADS=1" bash tools/gpu_ab.sh > $o/ab_big.txt 2>&1
mkdir -p /tmp/variants && mv mitsuba_amd/_build/libphip_*.so /tmp/variants/ 2>/dev/null
WORKLOADS="cornell 256;c42 256;c62 256;cmixed 256" AB_ENV="noflat3 PHIP_NO_FLAT3=1;noheads PHIP_NO_HEADS=1" bash tools/gpu_ab.sh > $o/ab_small.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "cornell or atrium or glass or zoo or work_list or record or mixed or textures or envmap or large_emitter" > $o/pytest.txt 2>&1
tail -3 $o/pytest.txt; cat $o/ab_big.txt $o/ab_small.txt
