#!/bin/bash
# round 6: forward-transform variant, two workgroups per block (shipped build) against the round-5 form (libfdc_amd_fwdold.so), same box, alternating;
# full band and (nearly) nothing written
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for lib in "" gr-fdc_amd/libfdc_amd_fwdold.so; do
    for args in "--force-path no-poly" "--force-path no-poly --sparse 1 --sparse-widths 256"; do
      FDC_AMD_LIB=${lib:+$PWD/$lib} python bench.py --config 2 --blocks 1024 --steps 100 --warmup 5 --no-cpu-baseline --no-end-to-end --timing-stride 1 --no-verify $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('${lib:-shipped(two-wg)}', '$args', d['ms_per_step'], r['kernel_ms_per_step'])"
    done
  done
done
