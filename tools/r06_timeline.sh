#!/bin/bash
# round 6: timeline of the look-ahead sink steps (who waits for whom): kernel trace of a short run, the last 1.5 ms printed
cd /tmp && export TMPDIR=/tmp
for cfg in 3 5; do
  rm -rf /tmp/tl$cfg
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl$cfg -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --payload device --lookahead --steps 12 --warmup 3 --no-cpu-baseline --no-end-to-end --no-kernel-timing > /tmp/tl$cfg.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/timeline.py /tmp/tl$cfg 1.5 > $GRAFT_REPO_ROOT/gpurun_out/timeline_cfg$cfg.txt 2>&1
  tail -3 /tmp/tl$cfg.log | cut -c1-300
done
