#!/bin/bash
# round 6: counter tables of the one-launch form of N = 4096 (k_f4096), configs[0]
cd $GRAFT_REPO_ROOT
PMC_TAG=f4096 bash profiles/pmc_run.sh f4096 --config 1 --no-end-to-end --no-verify --settle-ms 0 > gpurun_out/pmc_f4096.txt 2>&1
PMC_TAG=f4096 bash profiles/pmc_deep.sh f4096 --config 1 --no-end-to-end --no-verify --settle-ms 0 > gpurun_out/pmcd_f4096.txt 2>&1
rm -rf gpurun_out/pmc_*/pass* gpurun_out/pmcd_*/pass*
tail -40 gpurun_out/pmc_f4096.txt; tail -40 gpurun_out/pmcd_f4096.txt
