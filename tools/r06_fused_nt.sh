#!/bin/bash
# round 6: k_f4096 with streamed (nt) output stores (shipped) against plain stores (variant f4nt0): configs[0] R = 2 / 4, the N = 4096 plan-choice cases; parity first
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_fused4096_gpu.py -x -q 2>&1 | tail -2
for i in 1 2; do
  for tag in "" f4nt0; do
    lib=${tag:+$PWD/gr-fdc_amd/libfdc_amd_$tag.so}
    for args in "--relinvovl 2" "--relinvovl 4"; do
      FDC_AMD_LIB=$lib python bench.py --config 1 $args --steps 50 --warmup 5 --no-cpu-baseline --no-end-to-end --timing-stride 1 --no-verify 2>gpurun_out/fused_ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('${tag:-shipped (nt stores)}', '$args', d['ms_per_step'], list(r['kernel_ms_per_step'].values())[0])" || tail -3 gpurun_out/fused_ab.err
    done
  done
done
for tag in "" f4nt0; do
  lib=${tag:+$PWD/gr-fdc_amd/libfdc_amd_$tag.so}
  rm -f gpurun_out/pc_nt.txt
  FDC_AMD_LIB=$lib FDC_PLANCHOICE_LOG=gpurun_out/pc_nt.txt python -m pytest tests/test_plan_choice_gpu.py -q -k 4096 > /dev/null 2>&1
  echo "--- ${tag:-shipped (nt stores)}"; sed 's/ \[5.*//; s/; spectrum.*//' gpurun_out/pc_nt.txt | sort
done
timeout -k 10 600 python tools/fuzz_fused4096.py 100 4242 2>&1 | tail -1
