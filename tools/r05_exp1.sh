#!/bin/bash
# round 5, experiment 1 (GPU box): the memory side of stage 1 at configs[3], ablations of k_p1, counters
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r05_exp1; mkdir -p $O
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/ubench/p1_pattern.hip -o $O/p1_pattern 2>/dev/null
timeout -k 10 120 $O/p1_pattern > $O/p1_pattern.txt 2>&1; cat $O/p1_pattern.txt
B="python bench.py --config 4 --no-cpu-baseline --no-end-to-end --timing-stride 1 --steps 40 --warmup 5"
timeout -k 10 200 $B > $O/cfg4_default.json 2>$O/cfg4_default.err
for abl in 1 2; do
  FDC_DEBUG_ENV=1 FDC_ABLATE=$abl timeout -k 10 200 $B --no-verify > $O/cfg4_abl$abl.json 2>$O/cfg4_abl$abl.err
done
for nt in 0 1 2 11; do
  FDC_DEBUG_ENV=1 FDC_NT=$nt timeout -k 10 200 $B > $O/cfg4_nt$nt.json 2>$O/cfg4_nt$nt.err
done
timeout -k 10 200 $B --blocks 512 --chunk 512 > $O/cfg4_b512.json 2>$O/cfg4_b512.err
for f in $O/cfg4_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d['roofline']
    print(sys.argv[1].split('/')[-1], d['ms_per_step'], r['kernel_ms_per_step'], r['pipeline_frac'])
except Exception as e: print(sys.argv[1], 'failed', e)
PY
done
bash profiles/pmc_deep.sh r05_cfg4 --config 4 > $O/pmc_deep.log 2>&1; tail -70 $O/pmc_deep.log
