#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_exp6; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "split_plans or plan_classes" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
D=gr-fdc_amd/csrc/gr_blocks/blocks_demo
for args in "256 8192 0" "512 16384 0" "1024 16384 0" "64 4096 0" "256 1024 1"; do
  set -- $args
  timeout -k 10 120 $D stock 65536 2 256 $1 $2 $3 > $O/stock_$1_$3.json 2>$O/stock.err; cat $O/stock_$1_$3.json
done
timeout -k 10 200 python bench.py --two-widths --no-cpu-baseline --no-end-to-end --steps 40 --warmup 5 > $O/bench_two_widths.json 2>$O/err.txt; python -c "
import json; d=json.load(open('$O/bench_two_widths.json')); print(d['ms_per_step'], d['roofline']['pipeline_frac'], d['config']['kernel_plan'], d['verified']['max_rel_err'])"
timeout -k 10 200 python bench.py --mixed --no-cpu-baseline --no-end-to-end --steps 40 --warmup 5 > $O/bench_mixed.json 2>$O/err.txt; python -c "
import json; d=json.load(open('$O/bench_mixed.json')); print(d['ms_per_step'], d['roofline']['pipeline_frac'], d['config']['kernel_plan'], d['verified']['max_rel_err'])"
timeout -k 10 200 python bench.py --config 1 --no-cpu-baseline --no-end-to-end --steps 40 --warmup 5 > $O/bench_cfg1.json 2>$O/err.txt; python -c "
import json; d=json.load(open('$O/bench_cfg1.json')); print(d['ms_per_step'], d['roofline']['pipeline_frac'], d['config']['kernel_plan'], d['verified']['max_rel_err'])"
