#!/bin/bash
# usage: ab.sh <variant tag> [bench args...]; alternates shipped / variant three times on this box (same process count, same box)
TAG=$1; shift
for i in 1 2 3; do
  for lib in "" gr-fdc_amd/libfdc_amd_$TAG.so; do
    FDC_AMD_LIB=${lib:+$PWD/$lib} python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-end-to-end --timing-stride 1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('${lib:-shipped}', d['ms_per_step'], r['pipeline_frac'], r['kernel_ms_per_step'], d['verified']['max_rel_err'])"
  done
done
