#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_sinks_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "group_sums or hier or cfg3 or cfg5 or combined or pipelined" > gpurun_out/t_cells2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t_cells2.log
tail -3 gpurun_out/t_cells2.log
for i in 1 2; do
  for lib in "" gr-fdc_amd/libfdc_amd_fwdstg.so gr-fdc_amd/libfdc_amd_fw2.so gr-fdc_amd/libfdc_amd_fw2stg.so; do
    for args in "--force-path no-poly" "--force-path no-poly --sparse 1 --sparse-widths 256"; do
      FDC_AMD_LIB=${lib:+$PWD/$lib} python bench.py --config 2 --blocks 1024 --steps 100 --warmup 5 --no-cpu-baseline --no-end-to-end --timing-stride 1 $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('${lib:-shipped}', '$args', d['ms_per_step'], r['kernel_ms_per_step'], d['verified']['max_rel_err'])"
    done
  done
done
for cfg in 3 5; do
  python bench.py --config $cfg --payload device --lookahead --no-end-to-end --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg$cfg lookahead device', d['ms_per_step'], d['config']['blocks_per_step_per_gpu'], d['roofline']['kernel_ms_per_step'])"
done
bash tools/r06_census.sh
