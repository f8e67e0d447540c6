#!/bin/bash
# usage: tools/build_trace_variant.sh NAME "-DFLAG=1 ..." [file.hip ...]
# Builds gpu-raytracer_amd/csrc/_variants/NAME/libgrt_device.so: the default objects, with the listed translation units
# (default: kernels_trace.hip) recompiled with the extra flags. Select it at run time with GRT_DEVICE_LIB=<path>.
set -e
cd "$(dirname "$0")/../gpu-raytracer_amd"
NAME=$1; FLAGS=$2; shift 2 || true
FILES=${@:-kernels_trace.hip}
make -j16 csrc/libgrt_device.so > /dev/null
mkdir -p csrc/_variants/$NAME
OBJS=""
for src in csrc/*.hip; do
  base=$(basename $src .hip)
  if echo " $FILES " | grep -q " $base.hip "; then
    PERFILE=""; case $base in kernels_shade|kernels_post) PERFILE="-fno-slp-vectorize";; esac   # (the Makefile's HIPFLAGS_<unit>)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -I../include $PERFILE $FLAGS -c -o csrc/_variants/$NAME/$base.o $src
    OBJS="$OBJS csrc/_variants/$NAME/$base.o"
  else
    OBJS="$OBJS csrc/$base.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o csrc/_variants/$NAME/libgrt_device.so $OBJS
echo built csrc/_variants/$NAME/libgrt_device.so
