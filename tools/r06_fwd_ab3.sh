#!/bin/bash
cd $GRAFT_REPO_ROOT
FDC_AMD_LIB=$PWD/gr-fdc_amd/libfdc_amd_fw2deep.so python -m pytest tests/test_parity_gpu.py tests/test_sinks_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "spectrum or mixed or cfg3 or cfg5 or group_sums or part_of_the_band or split or hier" 2>&1 | tail -3
for i in 1 2; do
  for lib in "" gr-fdc_amd/libfdc_amd_fw2deep.so; do
    for args in "--force-path no-poly" "--force-path no-poly --sparse 1 --sparse-widths 256"; do
      FDC_AMD_LIB=${lib:+$PWD/$lib} python bench.py --config 2 --blocks 1024 --steps 100 --warmup 5 --no-cpu-baseline --no-end-to-end --timing-stride 1 $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('${lib:-shipped}', '$args', d['ms_per_step'], r['kernel_ms_per_step'], d['verified']['max_rel_err'])"
    done
  done
done
