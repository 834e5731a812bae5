#!/bin/bash
# round 6: k_f4096 with one block per workgroup where no row is wide (default) against two blocks everywhere (FDC_F4_TEAMS=2): the N = 4096 plan-choice cases, twice each
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_fused4096_gpu.py -x -q > gpurun_out/t_fused.log 2>&1; rc=$?; tail -3 gpurun_out/t_fused.log; [ $rc -eq 0 ] || exit $rc
rm -f gpurun_out/pc_t1.txt gpurun_out/pc_t2.txt gpurun_out/pc_t1w.txt
for i in 1 2; do
  FDC_PLANCHOICE_LOG=gpurun_out/pc_t1.txt python -m pytest tests/test_plan_choice_gpu.py -q -k 4096 > /dev/null 2>&1
  FDC_DEBUG_ENV=1 FDC_F4_TEAMS=2 FDC_PLANCHOICE_LOG=gpurun_out/pc_t2.txt python -m pytest tests/test_plan_choice_gpu.py -q -k 4096 > /dev/null 2>&1
  FDC_DEBUG_ENV=1 FDC_F4_TEAMS=1 FDC_PLANCHOICE_LOG=gpurun_out/pc_t1w.txt python -m pytest tests/test_plan_choice_gpu.py -q -k 4096 > /dev/null 2>&1
done
echo "--- default"; sed 's/; spectrum.*//' gpurun_out/pc_t1.txt | sort
echo "--- two blocks everywhere"; sed 's/; spectrum.*//' gpurun_out/pc_t2.txt | sort
echo "--- one block wherever it fits (wide rows too)"; sed 's/; spectrum.*//' gpurun_out/pc_t1w.txt | sort
