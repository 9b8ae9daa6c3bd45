#!/bin/bash
# round 4, call c: regeneration queue (product) / paired Wald tests / both profiled
set -x
mkdir -p gpurun_out/r4c
python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q -k "cornell or c2 or fused or fuzz or c1 or block_sizes or ragged or progressive or cancel or shards" 2>&1 | tail -5 > gpurun_out/r4c/pytest.txt
cat gpurun_out/r4c/pytest.txt
mv mitsuba_amd/_build/libphip_prof.so mitsuba_amd/_build/xprof.so; mv mitsuba_amd/_build/libphip_profpair.so mitsuba_amd/_build/xprofpair.so
WORKLOADS="cornell 256" bash tools/gpu_ab.sh 2>&1 | tee gpurun_out/r4c/ab.txt
PHIP_LIB=$PWD/mitsuba_amd/_build/xprof.so SPP=64 python tools/mega_profile.py gpurun_out/r4c/mega_profile_regenq.json
PHIP_LIB=$PWD/mitsuba_amd/_build/xprofpair.so SPP=64 python tools/mega_profile.py gpurun_out/r4c/mega_profile_regenq_pair.json
