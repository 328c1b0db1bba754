#!/bin/bash
# dev: build scripts/build/lib_<name>.so = the shipped objects with <files> recompiled under extra flags
# usage: scripts/dev/build_variant.sh <name> "<flags>" file1.hip [file2.hip ...]
set -e
name=$1; flags=$2; shift 2
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
C=$ROOT/ct-unet_amd/csrc
B=$C/build_$name; mkdir -p $B $ROOT/scripts/build
objs=""
for f in $C/*.hip; do
  b=$(basename $f .hip)
  if [[ " $* " == *" $b.hip "* ]]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -Wall -Wno-unused-function $flags -c $f -o $B/$b.o
    objs="$objs $B/$b.o"
  else
    objs="$objs $C/build/$b.o"
  fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs -ldl -o $ROOT/scripts/build/lib_$name.so
echo built lib_$name.so
