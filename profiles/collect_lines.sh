#!/bin/bash
# Every bench.py line of profiles/<round>/ again (no profiler passes): profiles/collect_lines.sh <tag>.  Writes gpurun_out/lines_<tag>/.
set -u
R=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/lines_$R
mkdir -p $OUT
cd $ROOT
run() { local name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > $OUT/$name.json 2> $OUT/$name.err || echo "$name failed"; echo "[lines] $name done"; }
run bench_default
run bench_default_driver_args --steps 20 --warmup 5
for c in 1 3 4 5; do run bench_cfg$c --config $c --steps 20 --warmup 3; done
for c in 3 5; do
  run bench_cfg${c}_device_payload --config $c --steps 20 --warmup 3 --no-cpu-baseline --payload device
  run bench_cfg${c}_host_engine --config $c --steps 20 --warmup 3 --no-cpu-baseline --sink-engine host --sync-sinks
done
run bench_one_ring_1024 --no-cpu-baseline --input-rings 1 --blocks 1024
run bench_offset37 --offset 37 --no-cpu-baseline
run bench_two_launch --no-cpu-baseline --force-path no-block
run bench_blocks4096 --blocks 4096 --chunk 4096 --steps 50 --warmup 5 --no-cpu-baseline
run bench_r4 --relinvovl 4 --no-cpu-baseline
run bench_mixed --mixed --no-cpu-baseline
run bench_sparse8 --sparse 8 --no-cpu-baseline
run bench_sparse8_full_spectrum --sparse 8 --no-cpu-baseline --force-path full-spectrum
run bench_cfg1_full_spectrum --config 1 --no-cpu-baseline --force-path full-spectrum
rm -f $OUT/*.err
ls $OUT | wc -l
