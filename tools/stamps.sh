#!/bin/bash
# The diagnostic build behind profiles/r05_headline_stamps.json: classify_uni_u5.hip with -DSHK_STAMPS=1 (s_memtime stamps at the phase
# boundaries of the three-pairs kernel, classify_uni.hpp) linked with the product's other objects into tools/variants/stamps.so.
#   bash tools/stamps.sh && gpurun -- 'bash tools/gpu_steps.sh st env:SHK_LIB_PATH=/root/repo/tools/variants/stamps.so py:tools/headline_stamps.py'
set -e
cd "$(dirname "$0")/../shark_amd/csrc"
make -j8 > /dev/null
mkdir -p ../../tools/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -Wno-unused-value -Wno-pass-failed -fno-gpu-rdc -DSHK_STAMPS=1 \
  -c classify_uni_u5.hip -o /tmp/classify_uni_u5_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/variants/stamps.so classify_uni_u2.o classify_uni_u3.o classify_uni_u4.o \
  /tmp/classify_uni_u5_stamps.o classify_uni_u6.o classify_uni_u8.o classify_uni_u10.o classify.o anchor_verdict.o index_build.o shark_hip.o device_scan.o device_sort.o measure.o -ldl
echo tools/variants/stamps.so
