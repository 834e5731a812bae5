#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_exp11; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_plan_choice_gpu.py -x -q -m gpu -k "512_bin or 1024_bin or banks_half or centred or two_widths or three_widths" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for w in 512 1024; do
  echo "== width $w"; bash tools/ab.sh oldplane --width $w 2>&1 | tee $O/ab_w$w.txt
  echo "== width $w R=4"; bash tools/ab.sh oldplane --width $w --relinvovl 4 2>&1 | tee $O/ab_w${w}_r4.txt
done
for w in 512 1024; do
  bash profiles/pmc_run.sh r05c_w$w --width $w > $O/pmc_w$w.log 2>&1; grep -E "^fdc|SQ_LDS_BANK_CONFLICT|SQ_LDS_IDX_ACTIVE|SQ_WAVE_CYCLES|SQ_ACTIVE_INST_VALU" gpurun_out/pmc_r05c_w$w/summary.txt
  cp gpurun_out/pmc_r05c_w$w/summary.txt $O/pmc_summary_w${w}_new.txt; rm -rf gpurun_out/pmc_r05c_w$w/pass*
done
