#!/bin/bash
# round 5: the mailbox variants of k_mega<MM_ALL> (-DMEGA_MAILBOX=1: wave 0 of a block serves the copper / glass vertices through two LDS mailboxes, no block barriers) on the mixed box
mkdir -p gpurun_out/r5mb; b=$PWD/mitsuba_amd/_build; o=gpurun_out/r5mb
for l in libphip libphip_mbv2 libphip_mbv3 libphip_mbv4 libphip; do PHIP_LIB=$b/$l.so SPP=256 timeout 120 python tools/gpu_scenes.py cmixed 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$l', d['Msamples/s'], d['kernel_ms']['fused_kernel_ms'])"; done 2>&1 | tee $o/mailbox_ab2.txt
PHIP_LIB=$b/libphip_mbdiag.so timeout 120 python - <<'PY' 2>&1 | tail -2 | tee $o/mailbox_diag.txt
from mitsuba_amd import _ffi, scene as S
from mitsuba_amd.integrator import Scene, PathHIP, HDRFilm
w=h=1024
sc=Scene(S.cornell_mixed(w,h,_ffi.gaussian_filter()).desc()); integ=PathHIP(maxDepth=-1); film=HDRFilm(w,h)
integ.render(sc, film, 64)
st=integ.stats.as_dict()
print({"deposited": st["closest_node_visits"], "client shaded a special itself": st["closest_triangle_tests"], "server withdrew": st["shadow_node_visits"], "server kept (R-box full)": st["shadow_triangle_tests"],
       "server passes": st["closest_rays"], "client wave-passes that looked for room": st["shadow_rays"], "client refills from the R-box": st["path_vertices"], "server lanes with a camera sample": st["samples"]})
PY
