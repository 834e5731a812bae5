#!/bin/bash
# round 6: the forward-transform variant after the two-workgroups-per-block change: parity suite, then the lines that contain it
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu -k "not plan_choice" > gpurun_out/t_all.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t_all.log
tail -4 gpurun_out/t_all.log
python bench.py --config 2 --blocks 1024 --force-path no-poly --no-end-to-end --no-cpu-baseline > gpurun_out/bench_fwd_full.json 2>gpurun_out/bench_fwd_full.err
python bench.py --config 2 --mixed --no-end-to-end --no-cpu-baseline > gpurun_out/bench_mixed.json 2>/dev/null
python bench.py --config 3 --payload device --lookahead --no-end-to-end --no-cpu-baseline > gpurun_out/bench_cfg3_dl.json 2>/dev/null
python bench.py --config 5 --payload device --lookahead --no-end-to-end --no-cpu-baseline > gpurun_out/bench_cfg5_dl.json 2>/dev/null
python - <<'PY'
import json
for f in ("bench_fwd_full","bench_mixed","bench_cfg3_dl","bench_cfg5_dl"):
    try:
        d=json.load(open("gpurun_out/%s.json"%f)); print(f, d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d.get("verified",{}).get("max_rel_err"))
    except Exception as e: print(f, "failed", e)
PY
