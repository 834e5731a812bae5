#!/bin/bash
# round 5, experiment 18: the spectrum path at N = 32768 / 16384 with the block kernel as its forward transform (k_blk256<P, ..., FWD>) against the two-pass
# transform it replaces there (--force-path no-block); mixed plan and a uniform bank sent down the spectrum path
O=gpurun_out/r05_exp18; mkdir -p $O
true
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-end-to-end"
run() { n=$1; shift; timeout -k 10 300 $B "$@" > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -5 $O/$n.err; return 1; }; python - "$O/$n.json" "$n" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=j["roofline"]
print("%-26s %.4f ms/step  pipeline_frac %s  kernels %s | %s" % (sys.argv[2], j["ms_per_step"], r.get("pipeline_frac"), r.get("kernel_ms_per_step"), j["config"].get("kernel_plan")))
PY
}
for NN in 32768 16384; do
b=$((134217728 / NN))
run mixed_n${NN} --mixed --blocklen $NN --channels $((NN / 256)) --blocks $b &&
run mixed_n${NN}_twopass --mixed --blocklen $NN --channels $((NN / 256)) --blocks $b --force-path no-block &&
run bank_n${NN}_spectrum --blocklen $NN --channels $((NN / 256)) --blocks $b --force-path no-poly || exit 1
done
