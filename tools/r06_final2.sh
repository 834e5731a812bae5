#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/final2; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1; echo "pytest rc=$?" >> $O/t_all.log; tail -3 $O/t_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
for f in FDC_NO_FUSED FDC_NO_POLY; do
  FDC_TEST_FORCE=$f python -m pytest tests -x -q -m gpu -k "not plan_choice" > $O/t_forced_$f.log 2>&1; echo "$f rc=$?"; tail -1 $O/t_forced_$f.log
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['verified']['max_rel_err'], d.get('end_to_end_h2d',{}).get('value'))"
timeout -k 10 300 python bench.py --config 1 --steps 20 --warmup 3 > $O/bench_cfg1.json 2> $O/bench_cfg1.err; python -c "
import json; d=json.load(open('$O/bench_cfg1.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['verified']['max_rel_err'], d.get('end_to_end_h2d',{}).get('value'))"
