#!/bin/bash
# round 6: k_f4096 with and without the idle waves' request for the successor's input (FDC_F4_PREFETCH: workgroups ahead, 0 = off), configs[0] R = 2 / 4; same box, three rounds
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_fused4096_gpu.py -x -q 2>&1 | tail -2
for i in 1 2 3; do
  for pf in -1 0 32 128; do
    for args in "--relinvovl 2" "--relinvovl 4"; do
      FDC_DEBUG_ENV=1 FDC_F4_PREFETCH=$pf python bench.py --config 1 $args --steps 50 --warmup 5 --no-cpu-baseline --no-end-to-end --timing-stride 1 --no-verify 2>gpurun_out/fused_pf.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('prefetch $pf', '$args', d['ms_per_step'], list(r['kernel_ms_per_step'].values())[0])" || tail -3 gpurun_out/fused_pf.err
    done
  done
done
