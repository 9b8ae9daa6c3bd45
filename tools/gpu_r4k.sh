#!/bin/bash
# round 4, call k: k_mega with the Wald tests dealt over the wave (MEGA_BALANCE) and the centre / half-extent box table (MEGA_FLAT_CH):
# parity of the product build, then the 2 x 2 A/B (variants built by tools/build_variant.sh: nobal, noch, base), then the phase profile
set -x
out=gpurun_out/r4k; mkdir -p $out
b=$PWD/mitsuba_amd/_build
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_round2.py tests/test_gpu_direct.py -m gpu -q --durations=12 2>&1 | tail -40 | tee $out/pytest.txt
for v in nobal noch; do
  PHIP_LIB=$b/libphip_$v.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "cornell or c1_config or block_sizes or ragged" 2>&1 | tail -3 | tee $out/pytest_$v.txt
done
WORKLOADS="cornell 256" REPEAT=3 bash tools/gpu_ab.sh 2>&1 | tee $out/ab.txt
PHIP_LIB=$b/xprof/libphip.so SPP=64 python tools/mega_profile.py $out/mega_profile.json 2>&1 | tail -1
PHIP_LIB=$b/xprof/libphip_base.so SPP=64 python tools/mega_profile.py $out/mega_profile_base.json 2>&1 | tail -1
