#!/bin/bash
# round 5, experiment 15: look-ahead form, compute units left to the decision chains
O=gpurun_out/r05_exp15; mkdir -p $O
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-end-to-end --payload device --lookahead"
run() { n=$1; shift; timeout -k 10 300 $B "$@" > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -5 $O/$n.err; return 1; }; python - "$O/$n.json" "$n" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c=j["config"]; nb=c["blocks_per_step_per_gpu"]
print("%-26s %4d blocks  %.4f ms/step = %.4f ms per 1024 blocks  %9.1f %s" % (sys.argv[2], nb, j["ms_per_step"], j["ms_per_step"]*1024/nb, j["value"], j["unit"]))
PY
}
for cfg in 5 3; do
for r in 12 16 24 32 48; do run cfg${cfg}_r$r --config $cfg --reserve-cus $r || exit 1; done
run cfg${cfg}_r16_2x --config $cfg --reserve-cus 16 --blocks 2048 || exit 1
done
