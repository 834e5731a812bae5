#!/bin/bash
# Builds gr-fdc_amd/libfdc_amd_<tag>.so with extra -D flags for every kernel file (A/B timing on one GPU box:
# FDC_AMD_LIB=<path> python bench.py ..., tools/ab.sh <tag> ...).  Usage: tools/build_variant.sh <tag> [-DNAME=VAL ...]
set -e
TAG=$1; shift
cd "$(dirname "$0")/../gr-fdc_amd/csrc"
OBJS=""
for f in fdc_kernels fdc_fast256 fdc_block256 fdc_block512 fdc_block1024 fdc_blocknarrow fdc_chanwide fdc_fused4096 fdc_sinks_dev; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result "$@" -c $f.hip -o /tmp/${f}_$TAG.o &
  OBJS="$OBJS /tmp/${f}_$TAG.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libfdc_amd_$TAG.so fdc_api.o fdc_sinks.o fdc_group.o $OBJS -Wl,-rpath,/opt/rocm/lib
echo built ../libfdc_amd_$TAG.so
