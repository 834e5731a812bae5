#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_sinks_gpu.py tests/test_fullsize_gpu.py tests/test_sinks_engines_gpu.py tests/test_sink_scenarios_gpu.py tests/test_sinks_group_gpu.py tests/test_cpp_blocks_gpu.py -x -q -m gpu > gpurun_out/t_sched.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t_sched.log
tail -3 gpurun_out/t_sched.log
for cfg in 3 5; do
  for form in "--payload device --lookahead" "--lookahead" "--payload device"; do
    python bench.py --config $cfg $form --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg$cfg', '$form', d['ms_per_step'], d['config']['blocks_per_step_per_gpu'], d['roofline']['kernel_ms_per_step'], [ (e['blocks_per_call'], e['value'], e['pdu_latency_calls']) for e in d.get('end_to_end_h2d', [])])"
  done
done
bash tools/r06_timeline.sh > /dev/null 2>&1
bash tools/r06_census.sh
