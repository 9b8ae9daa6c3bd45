timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -q -x -k "cornell or fused or cancel or progress or ld_sampler or empty" 2>&1 | tail -3
run() { echo "== $*"; env "$@" SPP=256 REPEAT=2 python tools/gpu_scenes.py cornell 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['scene'], d['Msamples/s'], d['kernel_ms']['fused_kernel_ms'], d['kernel_ms']['film_kernel_ms'], d['nodes/closest'], d['tris/closest'])"; }
run PHIP_NO_FLAT=1
run X=1
