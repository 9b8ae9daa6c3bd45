#!/bin/bash
# round 2, GPU call M: PHIP_SAMPLER_LD on the GPU (parity, drop-in with <sampler type="ldsampler">), perf sanity of the hot kernels
out=gpurun_out/r2m; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py tests/test_gpu_round2.py -m gpu -q -s -k "ld_sampler or plugin_inside or cornell_render or progressive" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
grep -E "passed|failed|rc=|^E |FAILED|ldsampler" $out/pytest.log | tail -30
for s in cornell atrium glass; do SPP=$([ $s = cornell ] && echo 256 || echo 64) python tools/gpu_scenes.py $s 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['scene'], d['Msamples/s'], d['kernel_ms'], d['iters'])"; done
