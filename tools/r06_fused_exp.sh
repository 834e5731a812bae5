#!/bin/bash
# round 6: what each stream's latency costs k_f4096 (variants built with tools/build_variant.sh f4eK -DF4_EXP=K), configs[0], same box, two rounds
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for tag in "" $(ls gr-fdc_amd/ | sed -n 's/^libfdc_amd_\(f4[a-z0-9]*\)\.so$/\1/p'); do
    lib=${tag:+$PWD/gr-fdc_amd/libfdc_amd_$tag.so}
    FDC_AMD_LIB=$lib python bench.py --config 1 --steps 50 --warmup 5 --no-cpu-baseline --no-end-to-end --timing-stride 1 --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('${tag:-shipped}', d['ms_per_step'], r['kernel_ms_per_step'])"
  done
done
