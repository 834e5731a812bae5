#!/bin/bash
# round 5, experiment 13: k_blk1024<P> and k_blknar<.., P> at N = 32768 / 16384 against the spectrum path, and the N = 65536 regression check
O=gpurun_out/r05_exp13; mkdir -p $O
B="python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-end-to-end"
run() { n=$1; shift; timeout -k 10 200 $B "$@" > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -3 $O/$n.err; return 1; }; python - "$O/$n.json" "$n" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=j["roofline"]; print("%-22s %.4f ms  %7.1f %s  frac %.4f  %s" % (sys.argv[2], j["ms_per_step"], j["value"], j["unit"], r["frac"], j["config"].get("kernel_plan")))
PY
}
for L in 1024 128 64; do
run w${L}_n65536 --width $L &&
run w${L}_n32768 --width $L --blocklen 32768 --blocks 4096 &&
run w${L}_n32768_spec --width $L --blocklen 32768 --blocks 4096 --force-path no-block &&
run w${L}_n16384 --width $L --blocklen 16384 --blocks 8192 &&
run w${L}_n16384_spec --width $L --blocklen 16384 --blocks 8192 --force-path no-block &&
run w${L}_n32768_r4 --width $L --blocklen 32768 --blocks 4096 --relinvovl 4 || exit 1
done
run w128_n16384_r4 --width 128 --blocklen 16384 --blocks 8192 --relinvovl 4
