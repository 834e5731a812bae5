#!/bin/bash
# The sink configurations (configs[2] / [4]) again after a change to the device engine: bench lines (payloads to the host, payloads
# left in HBM, host engine) and the rocprofv3 kernel summaries of steady-state steps.  Usage: profiles/collect_sinks.sh <tag>
set -u
R=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/sinks_$R
mkdir -p $OUT
cd $ROOT
for c in 3 5; do
  timeout -k 10 300 python3 bench.py --config $c --steps 20 --warmup 3 > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err || echo "bench cfg$c failed"
  timeout -k 10 300 python3 bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline --payload device > $OUT/bench_cfg${c}_device_payload.json 2> /dev/null || echo "bench cfg$c device payload failed"
  timeout -k 10 300 python3 bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline --sink-engine host --sync-sinks > $OUT/bench_cfg${c}_host_engine.json 2> /dev/null || echo "bench cfg$c host engine failed"
  echo "[collect_sinks] config $c lines done"
done
cd /tmp && export TMPDIR=/tmp
for c in 3 5; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg$c -- python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end --payload device --config $c > $OUT/stats_cfg$c.log 2>&1 || echo "rocprof cfg$c failed"
  f=$(find $OUT/stats_cfg$c -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then head -1 "$f" > $OUT/rocprof_kernel_stats_cfg$c.csv; grep "fdc::" "$f" >> $OUT/rocprof_kernel_stats_cfg$c.csv; fi
  rm -rf $OUT/stats_cfg$c
  echo "[collect_sinks] rocprof config $c done"
done
ls -la $OUT
