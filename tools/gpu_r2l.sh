#!/bin/bash
# round 2, GPU call L: ABI 5 (textured roughness / transmittance), box padding, the rest of the -m gpu suite
out=gpurun_out/r2l; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -q -s -k "not full_size_against_the_reference and not c2_at_full_size" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
grep -E "passed|failed|rc=|^E |FAILED|roughness|identical 0" $out/pytest.log | tail -40
for i in 1 2; do SPP=64 python tools/gpu_scenes.py atrium 2>&1 | tail -1 | cut -c1-400; done
SPP=128 python tools/gpu_scenes.py glass 2>&1 | tail -1 | cut -c1-400
