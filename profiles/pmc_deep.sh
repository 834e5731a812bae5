#!/bin/bash
# Second set of rocprofv3 counter passes (instruction issue, LDS queues, instruction cache) for bench.py; same rules as
# pmc_run.sh (one --pmc pass per set, nothing traced).  Usage: profiles/pmc_deep.sh <tag> [bench args...]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmcd_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAIT_INST_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQC_ICACHE_MISSES SQC_ICACHE_HITS SQC_ICACHE_REQ SQ_IFETCH SQ_IFETCH_LEVEL" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VALU2 SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d $OUT/pass$i -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing "$@" > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
  echo "pass $i done"
done
python3 $ROOT/profiles/pmc_summarize.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
