#!/bin/bash
# Builds an alternative libphip (same sources, extra -D flags) next to the product for A/B runs on the GPU box:
#   tools/build_variant.sh <tag> <flags...>   ->  mitsuba_amd/_build/libphip_<tag>.so   (load it with PHIP_LIB=...)
# Only phip.hip and phip_mega.hip are recompiled (the shading objects of the product build are reused).
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd); b=$root/mitsuba_amd/_build; c=$root/mitsuba_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $F "$@" -c $c/phip.hip -o $b/phip_$tag.o &
/opt/rocm/bin/hipcc $F "$@" -c $c/phip_mega.hip -o $b/phip_mega_$tag.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $b/libphip_$tag.so $b/phip_$tag.o $b/phip_mega_$tag.o $b/phip_shade0.o $b/phip_shade1.o $b/phip_shade2.o $b/phip_shade3.o -ldl
echo built $b/libphip_$tag.so
