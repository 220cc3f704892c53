#!/bin/bash
# register / scratch / LDS use of the kernels in an object file: tools/kernel_regs.sh shark_amd/csrc/classify.o [name-filter]
set -e
O=$1; F=${2:-classify_uni}
T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin $O $T/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/dev.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/dev.co > $T/notes.txt
python3 - $T/notes.txt "$F" <<'PY'
import re, sys, subprocess
t = open(sys.argv[1]).read()
for blk in t.split("- .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if sys.argv[2] not in name:
        continue
    g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem).replace("void shk::", "")
    print("%-58s vgpr %3s sgpr %3s spill v%s s%s scratch %4s lds %6s" % (dem, g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"),
          g("private_segment_fixed_size"), g("group_segment_fixed_size")))
PY
rm -rf $T
