#!/bin/bash
# VGPRs / spills / scratch of the kernels of one .hip file (device pass only).  Usage: tools/kres.sh gr-fdc_amd/csrc/fdc_block256.hip [pattern]
set -e
SRC=$1; PAT=${2:-k_}
T=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -c "$SRC" -o $T/dev.o -I$(dirname $SRC) ${KRES_FLAGS:-}
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --input=$T/dev.o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/dev.co > $T/notes.txt
python3 - $T/notes.txt "$PAT" <<'PY'
import re, subprocess, sys
t = open(sys.argv[1]).read()
for b in t.split("- .agpr_count")[1:]:
    name = re.search(r"\.name:\s+(\S+)", b)
    if not name:
        continue
    dn = subprocess.run(["/usr/bin/c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
    if sys.argv[2] in dn:
        g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, b).group(1)
        print("%-70s vgpr %s spill %s scratch %s sgpr %s" % (dn.split("(")[0][-70:], g("vgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"), g("sgpr_count")))
PY
rm -rf $T
