#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2; do
for lib in "" gr-fdc_amd/libfdc_amd_NO_PRIO.so gr-fdc_amd/libfdc_amd_NO_EAGER.so; do
for cfg in 3 5; do
  for form in "--payload device --lookahead" "--lookahead"; do
    FDC_AMD_LIB=${lib:+$PWD/$lib} python bench.py --config $cfg $form --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${lib:-shipped}', 'cfg$cfg', '$form', d['ms_per_step'], d['config']['blocks_per_step_per_gpu'])"
  done
done
done
done
