#!/bin/bash
# round 2, GPU call H: call-order parity stream on the GPU (parity / drop-in / round-2 tests), C2 at full size against the oracle
out=gpurun_out/r2h; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py tests/test_gpu_round2.py tests/test_gpu_direct.py -m gpu -x -q -s -k "not full_size_against_the_reference" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
grep -E "passed|failed|rc=|full size|pixel \(|bit-identical" $out/pytest.log | tail -40
