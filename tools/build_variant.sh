#!/bin/bash
# Builds a variant of the library with extra hipcc flags into variants/lib<name>.so (select it with RT_AMD_LIB=...).
# Usage: tools/build_variant.sh <name> [hipcc flags...]     e.g. tools/build_variant.sh cycles -DRT_OP_CYCLES
set -e
name=$1; shift
cd "$(dirname "$0")/../ray_tracing_amd/csrc"
mkdir -p ../../variants
/opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize --offload-arch=gfx950 "$@" -c rt_device.hip -o /tmp/rt_device_$name.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 host_scene.o host_bvh.o host_chap1.o host_cache.o host_fastbvh.o host_prep.o rt_viewer.o rt_hdrcache.o /tmp/rt_device_$name.o -o ../../variants/lib$name.so -L/usr/lib/gcc/x86_64-linux-gnu/11 -lgomp
echo built variants/lib$name.so
