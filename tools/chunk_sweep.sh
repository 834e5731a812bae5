#!/bin/bash
# GPU box helper: configs[3] shape (N = 262144, 1024 channels) at several launch-group sizes: does the stage-1 output G
# (1 MiB per block) staying in the memory-side cache between the two launches pay?  Prints step time and the summed kernel times.
for c in "$@"; do
  timeout -k 10 150 python bench.py --config 4 --no-cpu-baseline --chunk $c --timing-stride 1 > gpurun_out/chunk_$c.json 2>gpurun_out/chunk_$c.err || echo "chunk $c failed"
  python -c "import json;d=json.load(open('gpurun_out/chunk_$c.json'));r=d['roofline'];print('chunk',$c,'ms/step',d['ms_per_step'],'kernels',r['kernel_ms_per_step'],'frac',r['pipeline_frac'])"
done
