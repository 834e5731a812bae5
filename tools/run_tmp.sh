cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2
timeout -k 10 900 python -m pytest tests/test_sinks_engines_gpu.py tests/test_sink_scenarios_gpu.py tests/test_sinks_gpu.py -x -q -m gpu 2>&1 | tail -3
for c in 3 5; do for pl in host device; do
  python3 bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline --payload $pl > gpurun_out/s2/cfg${c}_$pl.json 2> gpurun_out/s2/cfg${c}_$pl.err
  python3 -c "
import json
d=json.load(open('gpurun_out/s2/cfg${c}_$pl.json'))
print('cfg$c $pl', d['ms_per_step'], d['value'], d['roofline']['pipeline_frac'])"
done; done
