#!/bin/bash
# call 7: k_shade with the emitter table addressed as LDS when only it fits (atrium), typed emitter-record loads, fire-and-forget wave statistics (product) vs HEAD (prev)
mkdir -p gpurun_out
b=$PWD/mitsuba_amd/_build
{
echo "== A/B big scenes"
WORKLOADS="atrium 64;glass 128" timeout 600 bash tools/gpu_ab.sh
echo "== cornell fused / wavefront: product, prev, product"
for m in 1 0; do for l in "" $b/libphip_prev.so ""; do PHIP_MEGA=$m PHIP_LIB=${l:-$b/libphip.so} SPP=256 REPEAT=2 python tools/gpu_scenes.py cornell 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('mega=$m', d['lib'][-20:], d['Msamples/s'], d['kernel_ms'], d['wall_ms'])"; done; done
echo "== parity"
timeout 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_direct.py -m gpu -x -q > gpurun_out/call7_pytest.txt 2>&1; grep -E "passed|failed|error" gpurun_out/call7_pytest.txt | tail -3
} > gpurun_out/call7.log 2>&1
