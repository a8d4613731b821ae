#!/bin/bash
# Build the product library from the working tree, then run a command on a GPU box:   bash tools/gpu.sh <timeout s> '<command>'
# (a stale libdistaff_hip.so travels silently otherwise: the snapshot takes whatever .so is in the tree)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
make -C "$root/distaff_amd/csrc" -j8 2>&1 | grep -E " error|Error " && exit 1
make -C "$root/oracle" > /dev/null
# the build stamp (git head, kernel table) travels with the snapshot: profile scripts on the box read it
(cd "$root" && python -c "import __graft_entry__ as g; g._build_info()" > /dev/null 2>&1) || echo "[gpu.sh] warning: _build_info() failed (kernel gate?)"
t=$1; shift
exec /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
