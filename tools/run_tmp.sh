cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/h1
run() { n=$1; shift; python3 bench.py --no-cpu-baseline --no-kernel-timing "$@" > gpurun_out/h1/$n.json 2> gpurun_out/h1/$n.err; python3 -c "
import json
d=json.load(open('gpurun_out/h1/$n.json'))
print('$n', d['ms_per_step'], d['value'], d['roofline']['pipeline_frac'])"; }
FDC_AMD_LIB=$PWD/gr-fdc_amd/libfdc_amd_st16.so timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "uniform_plan or relinvovl_4 or offsets or plan_classes or block_kernel" 2>&1 | tail -3
run base
FDC_AMD_LIB=$PWD/gr-fdc_amd/libfdc_amd_st16.so run st16
run base2
FDC_AMD_LIB=$PWD/gr-fdc_amd/libfdc_amd_st16.so run st16b
FDC_AMD_LIB=$PWD/gr-fdc_amd/libfdc_amd_st16.so run st16_r4 --relinvovl 4
run base_r4 --relinvovl 4
