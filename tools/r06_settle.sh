#!/bin/bash
# sustained against burst on one box: the driver's command (20 timed steps behind 150 ms), then 3000 timed steps (1 s), then the driver's command again
cd $GRAFT_REPO_ROOT
for k in 20 3000 20 6000 20; do
  python bench.py --steps $k --warmup 5 --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('lease', open('/proc/sys/kernel/random/boot_id').read().strip()[:8], 'steps', $k, 'ms_per_step', d['ms_per_step'], 'frac', r['frac'], 'pipeline_frac', r['pipeline_frac'])"
done
