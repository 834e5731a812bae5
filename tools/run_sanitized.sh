#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer run of the CPU-side code (build container; GPU sanitizers are not
# available on the pool):
#   * oracle/_san/libfdc_oracle.so            (make -C oracle SAN=1)
#   * gr-fdc_amd/_san/libfdc_amd.so           the C-ABI library with its HOST code instrumented (-Xarch_host -fsanitize=...)
#   * gr-fdc_amd/_san/libgnuradio-FDC-amd.so  the C++ gr::FDC block faces
# and then the -m "not gpu" test suite against them (argument validation, window design, parameter derivation, oracle
# against the fixtures, the gloo sharding test, the block faces' error paths).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
CLANG=/opt/rocm/lib/llvm/bin/clang
make -C oracle SAN=1 -s CC="$CLANG -shared-libsan"      # one sanitizer runtime for everything: clang's
mkdir -p gr-fdc_amd/_san /tmp/fdc_san
SANF="-fsanitize=address,undefined -fno-omit-frame-pointer"
cd gr-fdc_amd/csrc
for f in fdc_api fdc_kernels fdc_fast256 fdc_block256 fdc_block512 fdc_block1024 fdc_blocknarrow fdc_chanwide fdc_fused4096 fdc_sinks fdc_sinks_dev fdc_group; do
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -Xarch_host -fsanitize=address,undefined \
      -Xarch_host -fno-omit-frame-pointer -Wno-unused-result -c $f.hip -o /tmp/fdc_san/$f.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fsanitize=address,undefined -shared-libsan -o ../_san/libfdc_amd.so /tmp/fdc_san/*.o \
    -Wl,-rpath,/opt/rocm/lib -Wl,-soname,libfdc_amd.so
${CLANG}++ -shared-libsan -O1 -g -std=c++17 -fPIC -shared -Wall $SANF -o ../_san/libgnuradio-FDC-amd.so gr_blocks/fdc_blocks.cc -L../_san -lfdc_amd \
    -Wl,-rpath,'$ORIGIN' -Wl,-soname,libgnuradio-FDC-amd.so
cd "$ROOT"
RT="$($CLANG -print-file-name=libclang_rt.asan-x86_64.so)"
export LD_PRELOAD="$RT"
export LD_LIBRARY_PATH="$(dirname "$RT"):${LD_LIBRARY_PATH:-}"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:verify_asan_link_order=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
export FDC_ORACLE_LIB="$ROOT/oracle/_san/libfdc_oracle.so" FDC_AMD_LIB="$ROOT/gr-fdc_amd/_san/libfdc_amd.so"
python -m pytest tests/ -x -q -m "not gpu" "$@"
