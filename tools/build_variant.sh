#!/bin/bash
# Builds an alternative libphip (same sources, extra flags) next to the product for A/B runs on the GPU box:
#   [MAIN_FLAGS=..] [MEGA_FLAGS=..] [MEGAW_FLAGS=..] [SHADE_FLAGS=..] tools/build_variant.sh <tag> [flags for every unit...]
#   ->  mitsuba_amd/_build/libphip_<tag>.so   (load it with PHIP_LIB=...)
# Units whose flags equal the product's are not recompiled (the product's objects are linked).  Experiment builds (the measured alternatives of DESIGN.md 9:
# earlier kernel generations, algorithm-selecting environment variables) add -DPHIP_EXPERIMENTS=1 to every unit:  tools/build_variant.sh exp -DPHIP_EXPERIMENTS=1
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd); b=$root/mitsuba_amd/_build; c=$root/mitsuba_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function"
PROD_MEGA="-mllvm -disable-machine-licm"
objs=""; n=0
throttle() { n=$((n + 1)); if [ $((n % $(nproc))) -eq 0 ]; then wait; fi; }
for f in 11 3 2 1 8 0; do for q in 0 1 3 2; do
  if [ -n "$SHADE_FLAGS$*" ]; then /opt/rocm/bin/hipcc $F $SHADE_FLAGS "$@" -DSHADE_FEAT=$f -DSHADE_PART=$q -c $c/phip_shade.hip -o $b/phip_shade${f}_${q}_$tag.o & objs="$objs $b/phip_shade${f}_${q}_$tag.o"; throttle
  else objs="$objs $b/phip_shade${f}_$q.o"; fi
done; done
# phip_mega.hip is three objects: -DMEGA_PART=0 (scenes in LDS), -DMEGA_PART=1 (the 8-wide tree in memory), -DMEGA_PART=2 (`direct`); MEGAW_FLAGS / MEGAD_FLAGS = flags for the second / third one only
if [ -n "$MEGA_FLAGS$*" ]; then /opt/rocm/bin/hipcc $F ${MEGA_FLAGS:-$PROD_MEGA} "$@" -DMEGA_PART=0 -c $c/phip_mega.hip -o $b/phip_mega_$tag.o & objs="$objs $b/phip_mega_$tag.o"; throttle; else objs="$objs $b/phip_mega.o"; fi
if [ -n "$MEGA_FLAGS$MEGAW_FLAGS$*" ]; then /opt/rocm/bin/hipcc $F ${MEGA_FLAGS:-$PROD_MEGA} $MEGAW_FLAGS "$@" -DMEGA_PART=1 -c $c/phip_mega.hip -o $b/phip_megaw_$tag.o & objs="$objs $b/phip_megaw_$tag.o"; throttle; else objs="$objs $b/phip_megaw.o"; fi
if [ -n "$MEGA_FLAGS$MEGAD_FLAGS$*" ]; then /opt/rocm/bin/hipcc $F ${MEGA_FLAGS:-$PROD_MEGA} $MEGAD_FLAGS "$@" -DMEGA_PART=2 -c $c/phip_mega.hip -o $b/phip_megad_$tag.o & objs="$objs $b/phip_megad_$tag.o"; throttle; else objs="$objs $b/phip_megad.o"; fi
if [ -n "$MAIN_FLAGS$*" ]; then /opt/rocm/bin/hipcc $F $MAIN_FLAGS "$@" -DPHIP_BUILD_ID="\"variant-$tag\"" -c $c/phip.hip -o $b/phip_$tag.o & objs="$objs $b/phip_$tag.o"; else objs="$objs $b/phip.o"; fi
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $b/libphip_$tag.so $objs -ldl
echo built $b/libphip_$tag.so
