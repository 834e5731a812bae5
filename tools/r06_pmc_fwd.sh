#!/bin/bash
# round 6: counter tables of the forward-transform variant k_blk256<8,...,FWD>: full band, and with (nearly) nothing written
cd $GRAFT_REPO_ROOT
export PMC_CONFIG=2 PMC_BLOCKS=1024
PMC_TAG=fwd_full bash profiles/pmc_run.sh fwd_full --config 2 --blocks 1024 --force-path no-poly --no-end-to-end --no-verify > gpurun_out/pmc_fwd_full.txt 2>&1
PMC_TAG=fwd_full bash profiles/pmc_deep.sh fwd_full --config 2 --blocks 1024 --force-path no-poly --no-end-to-end --no-verify > gpurun_out/pmcd_fwd_full.txt 2>&1
PMC_TAG=fwd_none bash profiles/pmc_run.sh fwd_none --config 2 --blocks 1024 --sparse 1 --sparse-widths 256 --no-end-to-end --no-verify > gpurun_out/pmc_fwd_none.txt 2>&1
PMC_TAG=fwd_none bash profiles/pmc_deep.sh fwd_none --config 2 --blocks 1024 --sparse 1 --sparse-widths 256 --no-end-to-end --no-verify > gpurun_out/pmcd_fwd_none.txt 2>&1
python bench.py --config 2 --blocks 1024 --force-path no-poly --no-end-to-end --no-verify --no-cpu-baseline > gpurun_out/bench_fwd_full.json 2>/dev/null
python bench.py --config 2 --blocks 1024 --sparse 1 --sparse-widths 256 --no-end-to-end --no-verify --no-cpu-baseline > gpurun_out/bench_fwd_none.json 2>/dev/null
rm -rf gpurun_out/pmc_*/pass* gpurun_out/pmcd_*/pass*
tail -3 gpurun_out/pmc_fwd_full.txt
