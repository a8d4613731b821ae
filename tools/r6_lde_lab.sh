#!/bin/bash
# ON THE GPU BOX.  Alternates the product library with the laboratory builds under gpurun_tmp_libs/ (tools/ab_variant.sh) on ONE lease:
# device time of ntt_pass_a / ntt_pass_b per commit of 20 random columns at 2^20 and 2^22 (tools/r6_lde_lab.py).
#   bash tools/r6_lde_lab.sh [rounds] [sizes]      -> gpurun_out/r6_lde_lab.txt
ROUNDS=${1:-3}; SIZES=${2:-"20 22"}
mkdir -p gpurun_out; : > gpurun_out/r6_lde_lab.txt
for r in $(seq $ROUNDS); do
  for n in $SIZES; do
    python tools/r6_lde_lab.py product $n >> gpurun_out/r6_lde_lab.txt 2>/dev/null
    for v in $(ls gpurun_tmp_libs 2>/dev/null); do
      DISTAFF_HIP_LIB=gpurun_tmp_libs/$v/distaff_amd/libdistaff_hip.so python tools/r6_lde_lab.py $v $n >> gpurun_out/r6_lde_lab.txt 2>/dev/null
    done
  done
done
python - <<'PY'
import collections
rows = collections.defaultdict(list)
for line in open("gpurun_out/r6_lde_lab.txt"):
    p = line.split()
    if len(p) >= 5:
        rows[(p[0], p[1])].append([float(p[2]), float(p[3]), float(p[4]), float(p[6])])
print("%-18s %5s %3s %10s %10s %10s %10s" % ("library", "log_n", "n", "pass_a ms", "pass_b ms", "leaves ms", "lde phase"))
for (name, n), v in sorted(rows.items(), key=lambda kv: (kv[0][1], kv[0][0] != "product", kv[0][0])):
    mean = [sum(c) / len(v) for c in zip(*v)]
    print("%-18s %5s %3d %10.3f %10.3f %10.3f %10.3f   (pass_a min %.3f max %.3f)" % (name, n, len(v), mean[0], mean[1], mean[2], mean[3], min(r[0] for r in v), max(r[0] for r in v)))
PY
