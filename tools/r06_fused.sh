#!/bin/bash
# round 6: the one-launch form of N = 4096 (fdc_fused4096.hip): parity, then configs[0] with and without it at R = 2 and R = 4
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_fused4096_gpu.py -x -q > gpurun_out/t_fused.log 2>&1; rc=$?; tail -5 gpurun_out/t_fused.log
[ $rc -eq 0 ] || exit $rc
for r in 2 4; do
  for f in "" "--force-path no-fused"; do
    tag=$(echo "R${r}${f}" | tr -d ' -')
    timeout -k 10 300 python bench.py --config 1 --relinvovl $r $f --no-end-to-end --no-cpu-baseline > gpurun_out/bench_cfg1_$tag.json 2> gpurun_out/bench_cfg1_$tag.err || exit 1
    python - <<P
import json
d=json.loads(open("gpurun_out/bench_cfg1_$tag.json").read().strip().splitlines()[-1])
print("$tag", d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms_per_step"], d["config"]["kernel_plan"], d["verified"]["max_rel_err"])
P
  done
done
