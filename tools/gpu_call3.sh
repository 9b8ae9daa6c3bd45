#!/bin/bash
# call 3: k_rays_w block size / LDS node cache at 6 waves; k_mega with pinned wave-uniform values (product) vs HEAD (prev); k_shade<0> (Cornell, wavefront) before / after the one-round-trip head
mkdir -p gpurun_out
b=$PWD/mitsuba_amd/_build
{
echo "== A/B big scenes"
mv $b/libphip_preshade.so $b/preshade.keep
WORKLOADS="atrium 64;glass 128" timeout 600 bash tools/gpu_ab.sh
mv $b/preshade.keep $b/libphip_preshade.so
echo "== cornell fused: product (k_mega pinned), prev, product"
for l in "" $b/libphip_prev.so ""; do PHIP_LIB=${l:-$b/libphip.so} SPP=256 REPEAT=2 python tools/gpu_scenes.py cornell 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['lib'][-20:], d['Msamples/s'], d['kernel_ms'], d['wall_ms'])"; done
echo "== cornell wavefront: product, preshade, product"
for l in "" $b/libphip_preshade.so ""; do PHIP_MEGA=0 PHIP_LIB=${l:-$b/libphip.so} SPP=256 REPEAT=2 python tools/gpu_scenes.py cornell 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['lib'][-20:], d['Msamples/s'], d['kernel_ms'], d['wall_ms'])"; done
echo "== parity k_mega"
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cornell or c2_at_full" 2>&1 | tail -3
} > gpurun_out/call3.log 2>&1
