cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2
timeout -k 10 600 python -m pytest tests/test_sinks_engines_gpu.py tests/test_sink_scenarios_gpu.py tests/test_sinks_gpu.py -x -q -m gpu 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
for c in 5; do
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/s2/st$c -- python3 $GRAFT_REPO_ROOT/bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --payload device > $GRAFT_REPO_ROOT/gpurun_out/s2/log$c.txt 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/s2/st$c -name "*kernel_stats.csv" | head -1)
python3 - "$f" > $GRAFT_REPO_ROOT/gpurun_out/s2/kern$c.txt <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "fdc::" in r["Name"] or "rocclr" in r["Name"]:
        print("%-50s calls %5s avg %10.1f us total %8.2f ms" % (r["Name"][:50], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/s2/st$c
done
cd $GRAFT_REPO_ROOT
for c in 5; do for pl in host device; do
  python3 bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline --payload $pl > gpurun_out/s2/cfg${c}_$pl.json 2> gpurun_out/s2/cfg${c}_$pl.err
  python3 -c "
import json
d=json.load(open('gpurun_out/s2/cfg${c}_$pl.json'))
print('cfg$c $pl', d['ms_per_step'], d['value'], d['roofline']['pipeline_frac'])"
done; done
