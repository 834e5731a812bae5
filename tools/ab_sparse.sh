#!/bin/bash
# GPU box helper: plans that read part of the band only, with and without the partial spectrum write (FDC_PIPE_FULL_SPECTRUM)
for args in "--config 1" "--sparse 8" "--sparse 32"; do
  for fp in "" "--force-path full-spectrum"; do
    timeout -k 10 150 python bench.py $args $fp --no-cpu-baseline > gpurun_out/ab_sparse.json 2> gpurun_out/ab_sparse.err || echo "failed: $args $fp"
    python -c "import json;d=json.load(open('gpurun_out/ab_sparse.json'));r=d['roofline'];print('$args $fp:', d['ms_per_step'], 'ms/step', d['value'], 'Ms/s', r['kernel_ms_per_step'], 'frac', r['pipeline_frac'])"
  done
done
