#!/bin/bash
# Lines the default collection does not cover: the reference's default overlap (R = 4) at configs[1], a mixed-width
# 256-channel plan, and PMC traffic for configs other than the headline.  Usage: profiles/collect_extra.sh <tag> [what...]
#   what: r4 mixed twowidths n16k n32k n16kr4 n32kr4 w512 w512r4 w1024 w1024r4 w128 w64 w128r4 w64r4 shortw la3 la5 pmc1 pmc3 pmc4 pmc5   (default: all)
set -u
TAG=${1:-r03}; shift || true
WHAT=${*:-r4 mixed twowidths n16k n32k n16kr4 n32kr4 w512 w512r4 w1024 w1024r4 w128 w64 w128r4 w64r4 shortw la3 la5 pmc1 pmc3 pmc4 pmc5}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/extra_$TAG
mkdir -p $OUT
cd $ROOT
stats() {   # name, bench args...
  local name=$1; shift
  timeout -k 10 300 python3 bench.py --steps 50 --warmup 5 "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err || echo "bench $name failed"
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$name -- python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end "$@" > $OUT/stats_$name.log 2>&1 ) || echo "rocprof $name failed"
  local f=$(find $OUT/stats_$name -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then head -1 "$f" > $OUT/rocprof_kernel_stats_$name.csv; grep "fdc::" "$f" >> $OUT/rocprof_kernel_stats_$name.csv; fi
  rm -rf $OUT/stats_$name
}
pmc() {     # config, blocks, blocklen
  PMC_CONFIG=$1 PMC_BLOCKS=$2 PMC_BLOCKLEN=$3 bash profiles/pmc_run.sh ${TAG}_cfg$1 --config $1 > $OUT/pmc_cfg$1.log 2>&1
  cp gpurun_out/pmc_${TAG}_cfg$1/summary.txt $OUT/pmc_summary_cfg$1.txt 2>/dev/null
  cp gpurun_out/pmc_${TAG}_cfg$1/pmc_traffic.json $OUT/pmc_traffic_cfg$1.json 2>/dev/null
  rm -rf gpurun_out/pmc_${TAG}_cfg$1/pass*/
}
for w in $WHAT; do
  case $w in
    r4) stats r4 --relinvovl 4 --no-cpu-baseline ;;
    mixed) stats mixed --mixed --no-cpu-baseline ;;
    twowidths) stats two_widths --two-widths --no-cpu-baseline ;;    # round 5: a 256-bin bank + a 512-bin bank, two launches
    # the one-kernel form at the other block lengths (round 4): the same number of samples per step as the headline
    n16k) stats n16k --blocklen 16384 --channels 64 --blocks 8192 --no-cpu-baseline ;;
    n32k) stats n32k --blocklen 32768 --channels 128 --blocks 4096 --no-cpu-baseline ;;
    n16kr4) stats n16kr4 --blocklen 16384 --channels 64 --blocks 8192 --relinvovl 4 --no-cpu-baseline ;;
    n32kr4) stats n32kr4 --blocklen 32768 --channels 128 --blocks 4096 --relinvovl 4 --no-cpu-baseline ;;
    # uniform banks of other channel widths (round 4): the 512-bin block kernel, the narrow-channel block kernel (128 and 64 bins)
    w512) stats w512 --width 512 --no-cpu-baseline ;;
    w512r4) stats w512r4 --width 512 --relinvovl 4 --no-cpu-baseline ;;
    w1024) stats w1024 --width 1024 --no-cpu-baseline ;;
    w1024r4) stats w1024r4 --width 1024 --relinvovl 4 --no-cpu-baseline ;;
    w128) stats w128 --width 128 --no-cpu-baseline ;;
    w64) stats w64 --width 64 --no-cpu-baseline ;;
    w128r4) stats w128r4 --width 128 --relinvovl 4 --no-cpu-baseline ;;
    w64r4) stats w64r4 --width 64 --relinvovl 4 --no-cpu-baseline ;;
    # round 5: the other widths' block kernels at N = 32768 / 16384 (pass-count template); the sinks' look-ahead form with payloads in HBM
    shortw) for L in 512 1024 128 64; do stats w${L}_n32k --width $L --blocklen 32768 --blocks 4096 --no-cpu-baseline; stats w${L}_n16k --width $L --blocklen 16384 --blocks 8192 --no-cpu-baseline; done ;;
    la3) stats cfg3_lookahead --config 3 --payload device --lookahead --no-cpu-baseline ;;
    la5) stats cfg5_lookahead --config 5 --payload device --lookahead --no-cpu-baseline ;;
    pmc1) pmc 1 16384 4096 ;;
    pmc3) pmc 3 1024 65536 ;;
    pmc4) pmc 4 256 262144 ;;
    pmc5) pmc 5 1024 65536 ;;
  esac
  echo "$w done"
done
ls -la $OUT
