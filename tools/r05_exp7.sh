#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_exp7; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_group_gpu.py tests/test_cpp_blocks_gpu.py tests/test_fullsize_gpu.py -q -m gpu -k "host_path or state_carries or chunking or group or cpp or full_size_batch or hier or real_input or golden" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
D=gr-fdc_amd/csrc/gr_blocks/blocks_demo
for args in "64 4096 0" "128 8192 0" "256 8192 0" "512 16384 0" "256 1000 0 verify"; do
  set -- $args
  timeout -k 10 120 $D stock 65536 2 256 $1 $2 $3 ${4:-} > $O/stock_$1_$3${4:-}.json 2>$O/stock.err; cat $O/stock_$1_$3${4:-}.json; cat $O/stock.err
done
python tools/measure_extra.py 2>/dev/null | tail -12
