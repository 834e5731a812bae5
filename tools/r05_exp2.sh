#!/bin/bash
# round 5, experiment 2 (GPU box): k_p1d (two blocks ahead) parity + timing, ablations 3/4, stock-scheduler demo
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_exp2; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "two_stage or cfg4 or full_size_batch or short_calls or randomized or uniform_banks or chunking or cfg2_tiled" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
B="python bench.py --config 4 --no-cpu-baseline --no-end-to-end --timing-stride 1 --steps 40 --warmup 5"
for rep in 1 2; do
timeout -k 10 200 $B > $O/cfg4_deep_$rep.json 2>$O/err.txt
FDC_DEBUG_ENV=1 FDC_P1_DEEP=0 timeout -k 10 200 $B > $O/cfg4_shallow_$rep.json 2>$O/err.txt
done
for abl in 2 3 4; do
  FDC_DEBUG_ENV=1 FDC_ABLATE=$abl timeout -k 10 200 $B --no-verify > $O/cfg4_abl$abl.json 2>$O/err.txt
done
timeout -k 10 200 $B --blocks 512 --chunk 256 > $O/cfg4_b512c256.json 2>$O/err.txt
timeout -k 10 200 $B --blocks 512 --chunk 512 > $O/cfg4_b512c512.json 2>$O/err.txt
timeout -k 10 200 python bench.py --force-path no-block --no-cpu-baseline --no-end-to-end --timing-stride 1 --steps 40 --warmup 5 > $O/cfg2_twolaunch_deep.json 2>$O/err.txt
FDC_DEBUG_ENV=1 FDC_P1_DEEP=0 timeout -k 10 200 python bench.py --force-path no-block --no-cpu-baseline --no-end-to-end --timing-stride 1 --steps 40 --warmup 5 > $O/cfg2_twolaunch_shallow.json 2>$O/err.txt
for f in $O/cfg*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d['roofline']
    print(sys.argv[1].split('/')[-1], d['ms_per_step'], r['kernel_ms_per_step'], r['pipeline_frac'], d.get('verified',{}).get('max_rel_err'))
except Exception as e: print(sys.argv[1], 'failed', e)
PY
done
D=gr-fdc_amd/csrc/gr_blocks/blocks_demo
timeout -k 10 120 $D stock 65536 2 256 64 300 0 verify > $O/stock_verify.json 2>$O/stock_verify.err; cat $O/stock_verify.json $O/stock_verify.err
timeout -k 10 120 $D stock 65536 2 256 256 8192 > $O/stock_256.json 2>$O/stock.err; cat $O/stock_256.json $O/stock.err
timeout -k 10 120 $D stock 65536 2 256 256 8192 128 > $O/stock_128.json 2>$O/stock.err; cat $O/stock_128.json
timeout -k 10 120 $D stock 65536 2 256 64 4096 > $O/stock_64.json 2>$O/stock.err; cat $O/stock_64.json
timeout -k 10 120 $D stock 65536 2 256 256 512 1 > $O/stock_1.json 2>$O/stock.err; cat $O/stock_1.json
