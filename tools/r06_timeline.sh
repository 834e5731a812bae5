#!/bin/bash
# round 6: timeline of the look-ahead sink steps (who waits for whom): kernel trace of a short run, the last 1.5 ms printed;
# once with the library built without the priority streams and the eager chain (libfdc_amd_schedbefore.so), once as shipped
cd /tmp && export TMPDIR=/tmp
for tag in before shipped; do
  for cfg in 3 5; do
    rm -rf /tmp/tl$cfg
    if [ $tag = before ]; then export FDC_AMD_LIB=$GRAFT_REPO_ROOT/gr-fdc_amd/libfdc_amd_schedbefore.so; else unset FDC_AMD_LIB; fi
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl$cfg -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --payload device --lookahead --steps 12 --warmup 3 --no-cpu-baseline --no-end-to-end --no-kernel-timing > /tmp/tl$cfg.log 2>&1
    python3 $GRAFT_REPO_ROOT/tools/timeline.py /tmp/tl$cfg 1.5 > $GRAFT_REPO_ROOT/gpurun_out/timeline_cfg${cfg}_$tag.txt 2>&1
  done
done
ls -la $GRAFT_REPO_ROOT/gpurun_out/timeline_*
