#!/bin/bash
# Builds gr-fdc_amd/libfdc_amd_<tag>.so with extra -D flags for fdc_block256.hip, fdc_block512.hip, fdc_block1024.hip, fdc_blocknarrow.hip, fdc_kernels.hip and fdc_sinks_dev.hip (A/B timing on one GPU box:
# FDC_AMD_LIB=<path> python tools/sweep.py ...).  Usage: tools/build_variant.sh <tag> [-DNAME=VAL ...]
set -e
TAG=$1; shift
cd "$(dirname "$0")/../gr-fdc_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall "$@" -c fdc_block256.hip -o /tmp/fdc_block256_$TAG.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall "$@" -c fdc_block512.hip -o /tmp/fdc_block512_$TAG.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall "$@" -c fdc_block1024.hip -o /tmp/fdc_block1024_$TAG.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall "$@" -c fdc_blocknarrow.hip -o /tmp/fdc_blocknarrow_$TAG.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall "$@" -c fdc_kernels.hip -o /tmp/fdc_kernels_$TAG.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall "$@" -c fdc_sinks_dev.hip -o /tmp/fdc_sinks_dev_$TAG.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libfdc_amd_$TAG.so fdc_api.o /tmp/fdc_kernels_$TAG.o fdc_fast256.o /tmp/fdc_block256_$TAG.o /tmp/fdc_block512_$TAG.o /tmp/fdc_block1024_$TAG.o /tmp/fdc_blocknarrow_$TAG.o fdc_chanwide.o fdc_sinks.o /tmp/fdc_sinks_dev_$TAG.o fdc_group.o -Wl,-rpath,/opt/rocm/lib
echo built ../libfdc_amd_$TAG.so
