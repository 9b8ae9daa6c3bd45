#!/bin/bash
# Builds an alternative libphip (same sources, extra flags) next to the product for A/B runs on the GPU box:
#   [MAIN_FLAGS=..] [MEGA_FLAGS=..] [SHADE_FLAGS=..] tools/build_variant.sh <tag> [flags for every unit...]
#   ->  mitsuba_amd/_build/libphip_<tag>.so   (load it with PHIP_LIB=...)
# Units whose flags equal the product's are not recompiled (the product's objects are linked).
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd); b=$root/mitsuba_amd/_build; c=$root/mitsuba_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function"
PROD_MEGA="-mllvm -disable-machine-licm"
objs=""
if [ -n "$MAIN_FLAGS$*" ]; then /opt/rocm/bin/hipcc $F $MAIN_FLAGS "$@" -c $c/phip.hip -o $b/phip_$tag.o & objs="$objs $b/phip_$tag.o"; else objs="$objs $b/phip.o"; fi
if [ -n "$MEGA_FLAGS$*" ]; then /opt/rocm/bin/hipcc $F ${MEGA_FLAGS:-$PROD_MEGA} "$@" -c $c/phip_mega.hip -o $b/phip_mega_$tag.o & objs="$objs $b/phip_mega_$tag.o"; else objs="$objs $b/phip_mega.o"; fi
for f in 0 1 2 3 8 11; do
  if [ -n "$SHADE_FLAGS$*" ]; then /opt/rocm/bin/hipcc $F $SHADE_FLAGS "$@" -DSHADE_FEAT=$f -c $c/phip_shade.hip -o $b/phip_shade${f}_$tag.o & objs="$objs $b/phip_shade${f}_$tag.o"; else objs="$objs $b/phip_shade$f.o"; fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $b/libphip_$tag.so $objs -ldl
echo built $b/libphip_$tag.so
