#!/bin/bash
# build_variant.sh NAME FILE.hip [-Dflags...]: libvgkernels with ONE translation unit rebuilt under extra -D flags ->
# build/variants/libvg_NAME.so (load it with VG_KERNELS_SO; same-box A/B runs).  The other objects come from the default build.
set -e
cd "$(dirname "$0")/../../videoglamm_amd/csrc"
name=$1; src=$2; shift 2
V=../../build/variants; mkdir -p $V
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -Wno-unused-variable "$@" -c "$src" -o "$V/${src%.hip}_$name.o"
objs=""
for o in $(sed -n "s/^SRCS = //p" Makefile | sed "s/\.hip/.o/g; s/\.cpp/.o/g"); do
  if [ "$o" = "${src%.hip}.o" ]; then objs="$objs $V/${src%.hip}_$name.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o "$V/libvg_$name.so"
echo "built variants/libvg_$name.so"
