#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_fused4096_gpu.py -x -q > gpurun_out/t_fused.log 2>&1; rc=$?; tail -5 gpurun_out/t_fused.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tools/fuzz_fused4096.py 200 707 > gpurun_out/fuzz_fused4096_narrow.txt 2>&1; tail -3 gpurun_out/fuzz_fused4096_narrow.txt
