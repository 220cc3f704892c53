#!/bin/bash
# build an experimental variant of libsharkhip.so: tools/build_variant.sh NAME "-DSHK_...=.."   -> build/variants/NAME.so
# (used with SHK_LIB_PATH to A/B kernel choices on the GPU box; not part of the product build)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/tools/variants; mkdir -p $OUT/$NAME
cd $ROOT/shark_amd/csrc
for f in classify_uni_u10 classify_uni_u5 classify_uni_u8 classify_uni_u6 classify_uni_u4 classify_uni_u3 classify_uni_u2 classify anchor_verdict index_build shark_hip device_scan device_sort measure; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc "$@" -c $f.hip -o $OUT/$NAME/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/$NAME.so $OUT/$NAME/*.o -ldl
rm -rf $OUT/$NAME
echo built $OUT/$NAME.so
