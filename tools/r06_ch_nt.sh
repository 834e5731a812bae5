#!/bin/bash
# round 6: the channel kernels of the spectrum path (k_c256 / k_c512 / k_c1024) with streamed (nt) output stores (variant chnt, -DFDC_CH_NT=1) against plain stores
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for tag in "" c256nt; do
    lib=${tag:+$PWD/gr-fdc_amd/libfdc_amd_$tag.so}
    for args in "--config 2 --blocks 1024 --mixed" "--config 2 --blocks 1024 --force-path no-poly" "--config 1 --force-path no-fused" "--config 2 --blocks 1024 --sparse 8"; do
      FDC_AMD_LIB=$lib python bench.py $args --steps 50 --warmup 5 --no-cpu-baseline --no-end-to-end --timing-stride 1 --no-verify 2>gpurun_out/ch_nt.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('${tag:-working}', '$args', d['ms_per_step'], r['kernel_ms_per_step'])" || tail -3 gpurun_out/ch_nt.err
    done
  done
done
