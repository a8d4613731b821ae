#!/bin/bash
# Build the library of a git revision next to the working tree's, for A/B runs on ONE GPU box (box-to-box variance is ~2 %):
#   bash tools/ab_base.sh [rev]        -> gpurun_tmp_libs/base/distaff_amd/libdistaff_hip.so   (git-ignored, travels with gpurun)
#   DISTAFF_HIP_LIB=gpurun_tmp_libs/base/distaff_amd/libdistaff_hip.so python bench.py ...
set -e
rev=${1:-HEAD}
root=$(cd "$(dirname "$0")/.." && pwd)
rm -rf "$root/gpurun_tmp_libs/base" /tmp/ab_base_src
mkdir -p "$root/gpurun_tmp_libs/base" /tmp/ab_base_src
git -C "$root" archive "$rev" distaff_amd/csrc include | tar -x -C /tmp/ab_base_src
make -C /tmp/ab_base_src/distaff_amd/csrc -j8 > /dev/null
mkdir -p "$root/gpurun_tmp_libs/base/distaff_amd"
cp /tmp/ab_base_src/distaff_amd/libdistaff_hip.so "$root/gpurun_tmp_libs/base/distaff_amd/"
git -C "$root" rev-parse --short "$rev" > "$root/gpurun_tmp_libs/base/REV"
echo "base library of $(cat "$root/gpurun_tmp_libs/base/REV") built"
