#!/bin/bash
# Cross-compiles a VARIANT of the library next to the shipped one: tools/_variants/libbjxhip_<tag>.so
# usage: tools/build_variant.sh <tag> [extra hipcc flags, e.g. -DBJX_DENSE_PROBE] [-- file.hip ...]
# Only the listed .hip files (default: bjx_dense.hip) are recompiled with the flags; the other objects come from
# blackjax_amd/csrc/build/.  The variant travels to the GPU box with gpurun (git-ignored, not gpurun-ignored).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; shift
FLAGS=(); FILES=()
while [ $# -gt 0 ]; do
  if [ "$1" == "--" ]; then shift; FILES=("$@"); break; fi
  FLAGS+=("$1"); shift
done
[ ${#FILES[@]} -eq 0 ] && FILES=(bjx_dense.hip)
make -s -j8 -C $R/blackjax_amd/csrc
B=$R/tools/_variants/build_$TAG; mkdir -p $B
OBJS=()
for o in $R/blackjax_amd/csrc/build/*.o; do
  n=$(basename $o .o)
  skip=0; for f in "${FILES[@]}"; do [ "$n.hip" == "$f" ] && skip=1; done
  [ $skip -eq 0 ] && OBJS+=($o)
done
for f in "${FILES[@]}"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function "${FLAGS[@]}" \
     -c $R/blackjax_amd/csrc/$f -o $B/$(basename $f .hip).o
  OBJS+=($B/$(basename $f .hip).o)
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_variants/libbjxhip_$TAG.so "${OBJS[@]}"
echo $R/tools/_variants/libbjxhip_$TAG.so
