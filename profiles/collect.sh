#!/bin/bash
# Collects everything under profiles/<round>/ that bench.py's numbers are judged against, on the GPU box:
#   profiles/collect.sh r02
# 1. bench.py lines: default (configs[1]) and --config 1 / 3 / 4 / 5, plus an offset tiling; 2. rocprofv3 --kernel-trace
# --stats of the default command and of --config 4 (fdc:: rows of the kernel summary); 3. PMC passes (profiles/pmc_run.sh,
# profiles/pmc_deep.sh) of the default workload.  Writes to gpurun_out/collect_<round>/; copy what should be kept.
set -u
R=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/collect_$R
mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || echo "bench failed"
echo "[collect] default line done"
for c in 1 3 4 5; do
  timeout -k 10 300 python3 bench.py --config $c --steps 20 --warmup 3 > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err || echo "bench cfg$c failed"
done
echo "[collect] configs 1/3/4/5 done"
# the sink configs again with the payloads left in HBM (fdc_pdu.samples = device pointers), and on the host engine (round-2 form)
for c in 3 5; do
  timeout -k 10 300 python3 bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline --payload device > $OUT/bench_cfg${c}_device_payload.json 2> /dev/null || echo "bench cfg$c device payload failed"
  timeout -k 10 300 python3 bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline --sink-engine host --sync-sinks > $OUT/bench_cfg${c}_host_engine.json 2> /dev/null || echo "bench cfg$c host engine failed"
  # round 5: the look-ahead form (two spectrum buffers, the next batch's transform beside this batch's decisions), payloads in HBM and to the host
  timeout -k 10 300 python3 bench.py --config $c --steps 40 --warmup 5 --no-cpu-baseline --payload device --lookahead > $OUT/bench_cfg${c}_device_payload_lookahead.json 2> /dev/null || echo "bench cfg$c look-ahead failed"
  timeout -k 10 300 python3 bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline --lookahead > $OUT/bench_cfg${c}_lookahead.json 2> /dev/null || echo "bench cfg$c look-ahead (host payload) failed"
done
timeout -k 10 300 python3 bench.py --no-cpu-baseline --input-rings 1 --blocks 1024 > $OUT/bench_one_ring_1024.json 2> /dev/null || echo "bench one ring failed"
timeout -k 10 300 python3 bench.py --offset 37 --no-cpu-baseline > $OUT/bench_offset37.json 2> $OUT/bench_offset.err || echo "bench offset failed"
timeout -k 10 300 python3 bench.py --no-cpu-baseline --force-path no-block > $OUT/bench_two_launch.json 2> /dev/null || echo "bench two-launch failed"
# steps of 4096 blocks in one launch (and still three rings)
timeout -k 10 300 python3 bench.py --blocks 4096 --chunk 4096 --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_blocks4096.json 2> /dev/null || echo "bench 4096 failed"
echo "[collect] bench lines done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -- python3 $ROOT/bench.py --steps 200 --no-cpu-baseline --no-end-to-end > $OUT/stats_default.log 2>&1 || echo "rocprof default failed"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg4 -- python3 $ROOT/bench.py --steps 20 --no-cpu-baseline --no-end-to-end --config 4 > $OUT/stats_cfg4.log 2>&1 || echo "rocprof cfg4 failed"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg3 -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end --payload device --config 3 > $OUT/stats_cfg3.log 2>&1 || echo "rocprof cfg3 failed"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg5 -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end --payload device --config 5 > $OUT/stats_cfg5.log 2>&1 || echo "rocprof cfg5 failed"
echo "[collect] rocprof stats done"
for t in default cfg4 cfg3 cfg5; do
  f=$(find $OUT/stats_$t -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then head -1 "$f" > $OUT/rocprof_kernel_stats_$t.csv; grep "fdc::" "$f" >> $OUT/rocprof_kernel_stats_$t.csv; fi
done
cd $ROOT
PMC_BLOCKS=2048 bash profiles/pmc_run.sh ${R}_default > $OUT/pmc_default.log 2>&1
cp gpurun_out/pmc_${R}_default/summary.txt $OUT/pmc_summary_default.txt 2>/dev/null
cp gpurun_out/pmc_${R}_default/pmc_traffic.json $OUT/pmc_traffic.json 2>/dev/null
echo "[collect] pmc default done"
bash profiles/pmc_deep.sh ${R}_default > $OUT/pmc_deep.log 2>&1
cp gpurun_out/pmcd_${R}_default/summary.txt $OUT/pmc_summary_deep.txt 2>/dev/null
rm -rf $OUT/stats_default $OUT/stats_cfg4 $OUT/stats_cfg3 $OUT/stats_cfg5 gpurun_out/pmc_${R}_default/pass*/ gpurun_out/pmcd_${R}_default/pass*/
ls -la $OUT
