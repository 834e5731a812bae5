#!/bin/bash
# round 5: the whole GPU suite after the plan-selector refactor (banks of different widths), then the plan-choice timings
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_exp5; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu --deselect tests/test_plan_choice_gpu.py > $O/pytest_all.log 2>&1; tail -8 $O/pytest_all.log
rm -f $O/planchoice.txt; FDC_PLANCHOICE_LOG=$O/planchoice.txt timeout -k 10 600 python -m pytest tests/test_plan_choice_gpu.py -q -m gpu > $O/pytest_choice.log 2>&1; tail -15 $O/pytest_choice.log; cat $O/planchoice.txt
