#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_exp10; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "512_bin or 1024_bin or banks_half or centred" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for w in 512 1024; do
  echo "== width $w"; bash tools/ab.sh oldplane --width $w 2>&1 | tee $O/ab_w$w.txt
  echo "== width $w R=4"; bash tools/ab.sh oldplane --width $w --relinvovl 4 2>&1 | tee $O/ab_w${w}_r4.txt
done
for w in 512 1024; do
  bash profiles/pmc_run.sh r05b_w$w --width $w > $O/pmc_w$w.log 2>&1; grep -E "^fdc|SQ_LDS_BANK_CONFLICT|SQ_LDS_IDX_ACTIVE" gpurun_out/pmc_r05b_w$w/summary.txt
  cp gpurun_out/pmc_r05b_w$w/summary.txt $O/pmc_summary_w${w}_newplanes.txt; rm -rf gpurun_out/pmc_r05b_w$w/pass*
done
timeout -k 10 900 python tools/fuzz_paths.py 120 11 > $O/fuzz_r2.txt 2>&1; tail -3 $O/fuzz_r2.txt
timeout -k 10 600 python tools/fuzz_paths.py 60 12 4 > $O/fuzz_r4.txt 2>&1; tail -3 $O/fuzz_r4.txt
