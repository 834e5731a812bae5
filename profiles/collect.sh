#!/bin/bash
# Collects everything under profiles/<round>/ that bench.py's numbers are judged against, on the GPU box:
#   profiles/collect.sh r01
# 1. bench.py default line, 2. rocprofv3 --kernel-trace --stats of the same command (fdc:: rows of the kernel
# summary), 3. PMC passes (profiles/pmc_run.sh) for the default workload, 4. the same for configs[3]'s
# per-GPU shape (N=262144, 1024 channels).  Writes to gpurun_out/collect_<round>/; copy what should be kept.
set -u
R=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/collect_$R
mkdir -p $OUT
cd $ROOT
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || echo "bench failed"
python3 bench.py --blocklen 262144 --channels 1024 --blocks 256 --no-cpu-baseline > $OUT/bench_cfg4_shape.json 2> $OUT/bench_cfg4.err || echo "bench cfg4 failed"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -- python3 $ROOT/bench.py --steps 20 --no-cpu-baseline > $OUT/stats_default.log 2>&1 || echo "rocprof default failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg4 -- python3 $ROOT/bench.py --steps 20 --no-cpu-baseline --blocklen 262144 --channels 1024 --blocks 256 > $OUT/stats_cfg4.log 2>&1 || echo "rocprof cfg4 failed"
for t in default cfg4; do
  f=$(find $OUT/stats_$t -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then head -1 "$f" > $OUT/rocprof_kernel_stats_$t.csv; grep "fdc::" "$f" >> $OUT/rocprof_kernel_stats_$t.csv; fi
done
cd $ROOT
bash profiles/pmc_run.sh ${R}_default > $OUT/pmc_default.log 2>&1
cp gpurun_out/pmc_${R}_default/summary.txt $OUT/pmc_summary_default.txt 2>/dev/null
cp gpurun_out/pmc_${R}_default/pmc_traffic.json $OUT/pmc_traffic.json 2>/dev/null
rm -rf $OUT/stats_default $OUT/stats_cfg4
ls -la $OUT
