#!/bin/bash
# the GPU suite under each forced path (tests/conftest.py: FDC_TEST_FORCE); timing tests excluded
cd $GRAFT_REPO_ROOT
for f in FDC_NO_POLY FDC_NO_BLOCK FDC_FORCE_GENERIC; do
  FDC_TEST_FORCE=$f python -m pytest tests -x -q -m gpu -k "not plan_choice" > gpurun_out/t_forced_$f.log 2>&1; echo "$f rc=$?"; tail -2 gpurun_out/t_forced_$f.log
done
