#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/fuzz_final; mkdir -p $O
timeout -k 10 500 python tools/fuzz_paths.py 40 1234 > $O/fuzz_paths.txt 2>&1; tail -1 $O/fuzz_paths.txt
timeout -k 10 500 python tools/fuzz_sinks.py 60 1234 > $O/fuzz_sinks.txt 2>&1; tail -1 $O/fuzz_sinks.txt
timeout -k 10 500 python tools/fuzz_hier.py 60 1234 > $O/fuzz_hier.txt 2>&1; tail -1 $O/fuzz_hier.txt
