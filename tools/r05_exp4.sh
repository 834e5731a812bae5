#!/bin/bash
# round 5, experiment 4 (GPU box): manual waits (shipped) against the compiler's waits (variant libfdc_amd_autowait.so), same box
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_exp4; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "two_stage or cfg4 or full_size or short_calls or randomized or uniform_banks or chunking or cfg2_tiled or one_kernel_path or bench_launch or plan_classes or 256_bin_bank" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
echo "== headline"; bash tools/ab.sh autowait 2>&1 | tee $O/ab_default.txt
echo "== cfg4"; bash tools/ab.sh autowait --config 4 2>&1 | tee $O/ab_cfg4.txt
echo "== cfg2 two-launch"; bash tools/ab.sh autowait --force-path no-block 2>&1 | tee $O/ab_cfg2_twolaunch.txt
