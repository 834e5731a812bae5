#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_fullsize_gpu.py tests/test_parity_gpu.py -x -q -m gpu -k "262144 or cfg4 or 1024_slots or two_stage or uniform_plan" > gpurun_out/t_p2k.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t_p2k.log
tail -3 gpurun_out/t_p2k.log
bash tools/ab.sh p2knarrow --config 4
