#!/bin/bash
# Builds a tuning variant of the library: tools/build_variant.sh <name> [-DRT_...=v ...]  ->  build_variants/<name>.so
set -e
NAME=$1; shift
cd "$(dirname "$0")/../raytracing_amd/csrc"
make -s >/dev/null
mkdir -p ../../build_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fgpu-flush-denormals-to-zero -fno-slp-vectorize -fPIC -std=c++17 -I../../include -I../host -I. "$@" -c trace_kernels.hip -o /tmp/tk_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_variants/$NAME.so /tmp/tk_$NAME.o post_kernels.o rt_api.o host_*.o -lpthread -lz -ldl
echo built build_variants/$NAME.so
