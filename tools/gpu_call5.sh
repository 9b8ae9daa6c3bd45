#!/bin/bash
# call 5: regeneration with the tile origin kept in the info record (product) vs without (nohint); parity of the product
mkdir -p gpurun_out
b=$PWD/mitsuba_amd/_build
{
echo "== A/B big scenes"
WORKLOADS="atrium 64;glass 128" timeout 600 bash tools/gpu_ab.sh
echo "== cornell wavefront: product, nohint, product"
for l in "" $b/libphip_nohint.so ""; do PHIP_MEGA=0 PHIP_LIB=${l:-$b/libphip.so} SPP=256 REPEAT=2 python tools/gpu_scenes.py cornell 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['lib'][-20:], d['Msamples/s'], d['kernel_ms'], d['wall_ms'])"; done
echo "== parity"
timeout 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_direct.py tests/test_gpu_round2.py -m gpu -x -q 2>/dev/null | tail -5
} > gpurun_out/call5.log 2>&1
