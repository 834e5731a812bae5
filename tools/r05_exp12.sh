#!/bin/bash
# round 5, experiment 12: k_blk512<P> at N = 32768 / 16384 against the paths those banks ran before, and the N = 65536 regression check
O=gpurun_out/r05_exp12; mkdir -p $O
B="python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-end-to-end"
run() { n=$1; shift; timeout -k 10 200 $B "$@" > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -3 $O/$n.err; return 1; }; python - "$O/$n.json" "$n" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=j["roofline"]; print("%-22s %.4f ms  %.1f Gs/s  frac %.4f  kernel %s" % (sys.argv[2], j["ms_per_step"], j["value"]/1e9, r["frac"], j["config"].get("kernel_plan")))
PY
}
run w512_n65536 --width 512 &&
run w512_n32768 --width 512 --blocklen 32768 --blocks 4096 &&
run w512_n32768_noblock --width 512 --blocklen 32768 --blocks 4096 --force-path no-block &&
run w512_n32768_wide --width 512 --blocklen 32768 --blocks 4096 --force-path wide-uniform &&
run w512_n16384 --width 512 --blocklen 16384 --blocks 8192 &&
run w512_n16384_noblock --width 512 --blocklen 16384 --blocks 8192 --force-path no-block &&
run w512_n32768_r4 --width 512 --blocklen 32768 --blocks 4096 --relinvovl 4 &&
run w256_n32768 --blocklen 32768 --channels 128 --blocks 4096 &&
run w256_n16384 --blocklen 16384 --channels 64 --blocks 8192
