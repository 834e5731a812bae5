#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
D=gr-fdc_amd/csrc/gr_blocks/blocks_demo
for i in 1 2 3; do
  for g in 1 0; do
    echo -n "graded=$g 256: "; FDC_DEBUG_ENV=1 FDC_HOST_GRADED=$g timeout -k 10 120 $D stock 65536 2 256 256 8192 | python -c "import sys,json; d=json.load(sys.stdin); print(d['gsamples_per_s_in_work'])"
    echo -n "graded=$g 512: "; FDC_DEBUG_ENV=1 FDC_HOST_GRADED=$g timeout -k 10 120 $D stock 65536 2 256 512 16384 | python -c "import sys,json; d=json.load(sys.stdin); print(d['gsamples_per_s_in_work'])"
  done
done
