#!/bin/bash
# round 5: the bank kernels half a channel off their grid re-measured with the staged loads in (DESIGN 5.0's row predates them); counters of
# the bank kernels and of the two-launch kernels at N = 262144
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_exp9; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-end-to-end --steps 60 --warmup 5"
for w in 128 64 512 1024; do
  for r in 2 4; do
    timeout -k 10 200 $B --width $w --relinvovl $r > $O/bench_w${w}_r$r.json 2>$O/err.txt
    timeout -k 10 200 $B --width $w --relinvovl $r --offset $((w/2)) > $O/bench_w${w}_half_r$r.json 2>$O/err.txt
  done
done
for f in $O/bench_w*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d['roofline']
    print(sys.argv[1].split('/')[-1], d['ms_per_step'], r['pipeline_frac'], d['config']['kernel_plan'].split('path ')[1], d['verified']['max_rel_err'])
except Exception as e: print(sys.argv[1], 'failed', e)
PY
done
for w in 512 1024 64; do
  bash profiles/pmc_run.sh r05_w$w --width $w > $O/pmc_w$w.log 2>&1; cp gpurun_out/pmc_r05_w$w/summary.txt $O/pmc_summary_w$w.txt
  bash profiles/pmc_deep.sh r05_w$w --width $w > $O/pmcd_w$w.log 2>&1; cp gpurun_out/pmcd_r05_w$w/summary.txt $O/pmc_summary_deep_w$w.txt
  rm -rf gpurun_out/pmc_r05_w$w/pass* gpurun_out/pmcd_r05_w$w/pass*
  echo "pmc w$w done"
done
bash profiles/pmc_run.sh r05_cfg4 --config 4 > $O/pmc_cfg4.log 2>&1; cp gpurun_out/pmc_r05_cfg4/summary.txt $O/pmc_summary_cfg4.txt
bash profiles/pmc_deep.sh r05_cfg4 --config 4 > $O/pmcd_cfg4.log 2>&1; cp gpurun_out/pmcd_r05_cfg4/summary.txt $O/pmc_summary_deep_cfg4.txt
rm -rf gpurun_out/pmc_r05_cfg4/pass* gpurun_out/pmcd_r05_cfg4/pass*
echo done
