cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "relinvovl_4 or uniform_plan" 2>&1 | tail -8
python3 bench.py --no-cpu-baseline --relinvovl 4 --blocks 1024 > gpurun_out/r4/r4_1024.json 2> gpurun_out/r4/err.txt; python3 -c "
import json
d=json.load(open('gpurun_out/r4/r4_1024.json'))
print('R4 1024', d['ms_per_step'], d['value'], d['roofline']['pipeline_frac'], d['config']['kernel_path'])"
python3 bench.py --no-cpu-baseline --relinvovl 4 > gpurun_out/r4/r4_2048.json 2>> gpurun_out/r4/err.txt; python3 -c "
import json
d=json.load(open('gpurun_out/r4/r4_2048.json'))
print('R4 2048', d['ms_per_step'], d['value'], d['roofline']['pipeline_frac'], d['config']['kernel_path'])"
tail -3 gpurun_out/r4/err.txt
