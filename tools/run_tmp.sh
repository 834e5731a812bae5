cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/f1
timeout -k 10 900 python -m pytest tests/test_sinks_engines_gpu.py tests/test_sink_scenarios_gpu.py tests/test_sinks_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "not cfg2 and not cfg4" 2>&1 | tail -3
echo "[run] tests done"
timeout -k 10 300 python3 bench.py > gpurun_out/f1/bench_default.json 2> gpurun_out/f1/bench_default.err || echo "bench failed"
echo "[run] default bench done"
for c in 3 5; do for pl in host device; do
  timeout -k 10 300 python3 bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline --payload $pl > gpurun_out/f1/cfg${c}_$pl.json 2> gpurun_out/f1/cfg${c}_$pl.err
  python3 -c "
import json
d=json.load(open('gpurun_out/f1/cfg${c}_$pl.json'))
print('cfg$c $pl', d['ms_per_step'], d['value'], d['roofline']['pipeline_frac'])"
done; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/f1/st -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/f1/stats_default.log 2>&1 || echo "rocprof failed"
f=$(find $GRAFT_REPO_ROOT/gpurun_out/f1/st -name "*kernel_stats.csv" | head -1)
head -1 "$f" > $GRAFT_REPO_ROOT/gpurun_out/f1/rocprof_kernel_stats_default.csv; grep "fdc::" "$f" >> $GRAFT_REPO_ROOT/gpurun_out/f1/rocprof_kernel_stats_default.csv
rm -rf $GRAFT_REPO_ROOT/gpurun_out/f1/st
cat $GRAFT_REPO_ROOT/gpurun_out/f1/rocprof_kernel_stats_default.csv | cut -c1-60,380-460
tail -2 $GRAFT_REPO_ROOT/gpurun_out/f1/stats_default.log | cut -c1-400
