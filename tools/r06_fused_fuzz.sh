#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python tools/fuzz_fused4096.py 200 606 > gpurun_out/fuzz_fused4096_200cases.txt 2>&1; tail -3 gpurun_out/fuzz_fused4096_200cases.txt
FDC_TEST_FORCE=FDC_NO_FUSED timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "not plan_choice" > gpurun_out/t_forced_FDC_NO_FUSED.log 2>&1; echo "FDC_NO_FUSED rc=$?"; tail -2 gpurun_out/t_forced_FDC_NO_FUSED.log
