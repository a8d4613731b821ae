#!/bin/bash
# On the GPU box: alternate the short bench between the product library and the variant libraries under gpurun_tmp_libs/ for several
# rounds (clocks drift with temperature: single back-to-back runs differ by 2-3 %), then print the per-library means.
#   bash tools/ab_multi.sh [rounds] [variant names ...]      (default: 3 rounds, every variant present)
ROUNDS=${1:-3}; shift
VARS="$@"; [ -z "$VARS" ] && VARS=$(ls gpurun_tmp_libs 2>/dev/null)
mkdir -p gpurun_out; : > gpurun_out/ab_multi.txt
one() {   # name, env
    env $2 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-upload-leg --no-verify --allow-lib-override 2>&1 | grep "^{" | python -c "
import json, sys
b = json.loads(sys.stdin.read()); k = b['kernels']
air = sum(v['ms_per_step'] for n, v in k.items() if n.startswith('air_kernel'))
print('$1', b['ms_per_step'], b['phase_ms']['lde'], k.get('ntt_pass_a', {}).get('ms_per_step', 0), k.get('ntt_pass_b', {}).get('ms_per_step', 0), air, b['phase_ms']['trace_merkle'], b['phase_ms']['fri'])" >> gpurun_out/ab_multi.txt
}
for r in $(seq $ROUNDS); do
    one product "X_=1"
    for v in $VARS; do one $v "DISTAFF_HIP_LIB=gpurun_tmp_libs/$v/distaff_amd/libdistaff_hip.so"; done
done
python - <<'PY'
import collections
rows = collections.defaultdict(list)
for line in open("gpurun_out/ab_multi.txt"):
    p = line.split()
    rows[p[0]].append([float(x) for x in p[1:]])
print("%-14s %3s %9s %9s %9s %9s %9s %9s %9s" % ("library", "n", "proof ms", "lde", "pass_a", "pass_b", "air", "merkle", "fri"))
for name, v in rows.items():
    n = len(v)
    mean = [sum(c) / n for c in zip(*v)]
    print("%-14s %3d " % (name, n) + " ".join("%9.3f" % x for x in mean) + "   (proof min %.3f max %.3f)" % (min(r[0] for r in v), max(r[0] for r in v)))
PY
