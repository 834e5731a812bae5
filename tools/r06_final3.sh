#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/final3; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1; echo "pytest rc=$?" >> $O/t_all.log; tail -3 $O/t_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout -k 10 600 python tools/fuzz_fused4096.py 200 808 > $O/fuzz_fused4096.txt 2>&1; tail -1 $O/fuzz_fused4096.txt
FDC_DEBUG_ENV=1 FDC_F4_TEAMS=2 timeout -k 10 600 python tools/fuzz_fused4096.py 100 909 > $O/fuzz_fused4096_t2.txt 2>&1; tail -1 $O/fuzz_fused4096_t2.txt
FDC_PLANCHOICE_LOG=$O/plan_choice_4096.txt python -m pytest tests/test_plan_choice_gpu.py -q -k 4096 > $O/t_pc.log 2>&1; tail -1 $O/t_pc.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['verified']['max_rel_err'], d.get('end_to_end_h2d',{}).get('value'))"
timeout -k 10 300 python bench.py --config 1 --steps 20 --warmup 3 > $O/bench_cfg1.json 2> $O/bench_cfg1.err; python -c "
import json; d=json.load(open('$O/bench_cfg1.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['verified']['max_rel_err'], d.get('end_to_end_h2d',{}).get('value'))"
