#!/bin/bash
# GPU box helper: the detection tracker at other slab lengths (libfdc_amd_slab<L>.so from tools/build_variant.sh slab<L> -DFDC_DET_SLAB=<L>):
# the randomised engine comparison, then configs[4] with the payloads left in HBM
for L in "$@"; do
  export FDC_AMD_LIB=$PWD/gr-fdc_amd/libfdc_amd_slab$L.so
  timeout -k 10 200 python tools/fuzz_sinks.py 40 71 2>&1 | tail -1
  timeout -k 10 100 python bench.py --config 5 --no-cpu-baseline --payload device | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('slab $L:', d['ms_per_step'], 'ms per step')"
done
