#!/bin/bash
# round 6: what the notes quote for the one-launch form of N = 4096: suite, bench lines of configs[0] (with and without, R = 2 and 4), rocprof kernel stats, counter passes
cd $GRAFT_REPO_ROOT
O=gpurun_out/f4096; mkdir -p $O
echo "(suite: run separately)"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout -k 10 300 python bench.py --config 1 --steps 20 --warmup 3 > $O/bench_cfg1.json 2> $O/bench_cfg1.err
timeout -k 10 300 python bench.py --config 1 --steps 20 --warmup 3 --force-path no-fused --no-cpu-baseline > $O/bench_cfg1_two_launches.json 2>/dev/null
timeout -k 10 300 python bench.py --config 1 --steps 20 --warmup 3 --relinvovl 4 --no-cpu-baseline > $O/bench_cfg1_R4.json 2>/dev/null
timeout -k 10 300 python bench.py --config 1 --steps 20 --warmup 3 --relinvovl 4 --force-path no-fused --no-cpu-baseline > $O/bench_cfg1_R4_two_launches.json 2>/dev/null
for f in bench_cfg1 bench_cfg1_two_launches bench_cfg1_R4 bench_cfg1_R4_two_launches; do python -c "
import json; d=json.load(open('$O/$f.json')); r=d['roofline']; print('$f', d['value'], d['ms_per_step'], r['frac'], r['traffic'], d['verified']['max_rel_err'], (d.get('end_to_end_h2d') or {}).get('value'))"; done
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_cfg1 -- python3 $GRAFT_REPO_ROOT/bench.py --config 1 --steps 20 --no-cpu-baseline --no-end-to-end > $GRAFT_REPO_ROOT/$O/stats_cfg1.log 2>&1 )
f=$(find $O/stats_cfg1 -name "*kernel_stats.csv" | head -1); if [ -n "$f" ]; then head -1 "$f" > $O/rocprof_kernel_stats_cfg1.csv; grep "fdc::" "$f" >> $O/rocprof_kernel_stats_cfg1.csv; fi; rm -rf $O/stats_cfg1
cat $O/rocprof_kernel_stats_cfg1.csv
PMC_CONFIG=1 PMC_BLOCKS=16384 PMC_BLOCKLEN=4096 bash profiles/pmc_run.sh f4096 --config 1 --no-end-to-end --no-verify --settle-ms 0 > $O/pmc.log 2>&1
cp gpurun_out/pmc_f4096/summary.txt $O/pmc_summary_f4096.txt; cp gpurun_out/pmc_f4096/pmc_traffic.json $O/pmc_traffic_f4096.json
rm -rf gpurun_out/pmc_f4096/pass*
cat $O/pmc_traffic_f4096.json | head -20
