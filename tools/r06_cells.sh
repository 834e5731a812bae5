#!/bin/bash
# round 6: power cells from the forward kernel's group sums: parity, then configs[2] / [4] with and without (same box)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_sinks_gpu.py tests/test_fullsize_gpu.py tests/test_sinks_engines_gpu.py tests/test_parity_gpu.py -x -q -m gpu -k "group_sums or hier or cfg3 or cfg5 or combined or overlap_save or pipelined or spectrum" > gpurun_out/t_cells.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t_cells.log
tail -5 gpurun_out/t_cells.log
for cfg in 3 5; do
  for extra in "" "--no-fused-cells"; do
    for form in "--payload device --lookahead" "--payload device" ""; do
      python bench.py --config $cfg $form $extra --no-end-to-end --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg$cfg', '$form', '$extra', d['ms_per_step'], d['config']['blocks_per_step_per_gpu'], d['roofline']['kernel_ms_per_step'], d['config']['pdus_per_step'])"
    done
  done
done
