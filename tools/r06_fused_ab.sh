#!/bin/bash
# round 6: k_f4096 variants (gr-fdc_amd/libfdc_amd_f4*.so, tools/build_variant.sh) against the working build, configs[0] R = 2 / 4 and a full 256-bin band; same box, two rounds
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_fused4096_gpu.py -x -q 2>&1 | tail -2
for i in 1 2; do
  for tag in "" $(ls gr-fdc_amd/ | sed -n 's/^libfdc_amd_\(f4[a-z0-9]*\)\.so$/\1/p'); do
    lib=${tag:+$PWD/gr-fdc_amd/libfdc_amd_$tag.so}
    for args in "--relinvovl 2" "--relinvovl 4"; do
      FDC_AMD_LIB=$lib python bench.py --config 1 $args --steps 50 --warmup 5 --no-cpu-baseline --no-end-to-end --timing-stride 1 --no-verify 2>gpurun_out/fused_ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('${tag:-working}', '$args', d['ms_per_step'], list(r['kernel_ms_per_step'].values())[0])" || tail -3 gpurun_out/fused_ab.err
    done
  done
done
