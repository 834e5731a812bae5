#!/bin/bash
# the GPU suite under each forced path (tests/conftest.py: FDC_TEST_FORCE; the defaults are restored after every test); timing tests excluded; every failure listed
cd $GRAFT_REPO_ROOT
for f in ${@:-FDC_NO_FUSED FDC_NO_POLY FDC_NO_BLOCK FDC_FORCE_GENERIC}; do
  FDC_TEST_FORCE=$f timeout -k 10 900 python -m pytest tests -q -m gpu -k "not plan_choice" > gpurun_out/t_forced_$f.log 2>&1; echo "$f rc=$?"; grep "^FAILED" gpurun_out/t_forced_$f.log | cut -c1-150; tail -1 gpurun_out/t_forced_$f.log
done
