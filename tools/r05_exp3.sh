#!/bin/bash
# round 5, experiment 3 (GPU box): k_p1 with its row loads as inline assembly and the wait stated behind the stores
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_exp3; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "two_stage or cfg4 or full_size_batch or short_calls or randomized or uniform_banks or chunking or cfg2_tiled" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
B="python bench.py --config 4 --no-cpu-baseline --no-end-to-end --timing-stride 1 --steps 40 --warmup 5"
for rep in 1 2 3; do timeout -k 10 200 $B > $O/cfg4_$rep.json 2>$O/err.txt; done
for abl in 2 3 4; do
  FDC_DEBUG_ENV=1 FDC_ABLATE=$abl timeout -k 10 200 $B --no-verify > $O/cfg4_abl$abl.json 2>$O/err.txt
done
timeout -k 10 200 python bench.py --force-path no-block --no-cpu-baseline --no-end-to-end --timing-stride 1 --steps 40 --warmup 5 > $O/cfg2_twolaunch.json 2>$O/err.txt
timeout -k 10 200 python bench.py --no-cpu-baseline --no-end-to-end --steps 40 --warmup 5 > $O/cfg2_default.json 2>$O/err.txt
for f in $O/cfg*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d['roofline']
    print(sys.argv[1].split('/')[-1], d['ms_per_step'], r['kernel_ms_per_step'], r['pipeline_frac'], d.get('verified',{}).get('max_rel_err'))
except Exception as e: print(sys.argv[1], 'failed', e)
PY
done
