#!/bin/bash
# round 5, experiment 14: the sink configurations with and without the look-ahead form (payloads in HBM and to the host)
O=gpurun_out/r05_exp14; mkdir -p $O
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-end-to-end"
run() { n=$1; shift; timeout -k 10 300 $B "$@" > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -5 $O/$n.err; return 1; }; python - "$O/$n.json" "$n" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c=j["config"]; nb=c["blocks_per_step_per_gpu"]
print("%-26s %4d blocks  %.4f ms/step = %.4f ms per 1024 blocks  %9.1f %s  pdus/step %s" % (sys.argv[2], nb, j["ms_per_step"], j["ms_per_step"]*1024/nb, j["value"], j["unit"], c.get("pdus_per_step")))
PY
}
timeout -k 10 200 python -m pytest tests/test_sinks_gpu.py -x -q -m gpu 2>&1 | tail -2
for cfg in ${CFGS:-5 3}; do
run cfg${cfg}_dev --config $cfg --payload device &&
run cfg${cfg}_dev_la --config $cfg --payload device --lookahead &&
run cfg${cfg}_dev_la0 --config $cfg --payload device --lookahead --reserve-cus 0 &&
run cfg${cfg}_dev_la16 --config $cfg --payload device --lookahead --reserve-cus 16 &&
run cfg${cfg}_host --config $cfg &&
run cfg${cfg}_host_la --config $cfg --lookahead || exit 1
done
