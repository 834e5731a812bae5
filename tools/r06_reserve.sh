#!/bin/bash
# how many compute units the persistent forward kernel leaves to the bank's kernels (look-ahead form): throughput against --reserve-cus
cd $GRAFT_REPO_ROOT
for cfg in 3 5; do
  for r in 16 32 48 64 80; do
    python bench.py --config $cfg --payload device --lookahead --reserve-cus $r --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); nb=d['config']['blocks_per_step_per_gpu']; print('cfg$cfg reserve $r', 'blocks', nb, 'ms', d['ms_per_step'], 'us/block', round(1e3*d['ms_per_step']/nb,4), 'Msps', d['value'])"
  done
done
