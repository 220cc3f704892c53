#!/bin/bash
# local wrapper: rebuild everything that travels prebuilt, then send the command to the GPU box
# usage: tools/gpu.sh <timeout-seconds> '<command>'
set -e
cd "$(dirname "$0")/.."
make -C shark_amd/csrc -j8 all 2>&1 | grep -E "error|warning: unused|Error" || true
make -C shark_amd/csrc -q all || { echo "build is not up to date"; exit 1; }
make -C oracle all > /dev/null
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
