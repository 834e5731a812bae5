#!/bin/bash
# one lease = one line: the default bench as the driver runs it, appended to gpurun_out/boxes.txt by the caller
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import sys,json,socket; d=json.loads(sys.stdin.read()); r=d['roofline']
print('lease', open('/proc/sys/kernel/random/boot_id').read().strip()[:8], 'ms_per_step', d['ms_per_step'], 'frac', r['frac'], 'pipeline_frac', r['pipeline_frac'], 'Msps', d['value'], 'verified', d['verified']['max_rel_err'])"
