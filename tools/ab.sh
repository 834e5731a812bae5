#!/bin/bash
# usage: ab.sh <variant tag> [bench args...]; alternates shipped / variant three times on this box
TAG=$1; shift
for i in 1 2 3; do
  for lib in "" gr-fdc_amd/libfdc_amd_$TAG.so; do
    FDC_AMD_LIB=${lib:+$PWD/$lib} python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-end-to-end "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${lib:-shipped}', d['ms_per_step'], d['roofline']['pipeline_frac'], d['verified']['max_rel_err'])"
  done
done
