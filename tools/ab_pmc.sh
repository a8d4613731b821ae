#!/bin/bash
# On the GPU box: time + FETCH_SIZE / WRITE_SIZE of the transform kernels under the given environment settings (one counter pass each).
#   bash tools/ab_pmc.sh "" "DISTAFF_NTT_ORDER=1" ...
export TMPDIR=/tmp
R=$PWD
for e in "$@"; do
    for ctr in FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/abp; mkdir -p /tmp/abp
        (cd /tmp && env ${e:-X_=1} rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/abp -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-upload-leg --no-verify --allow-lib-override > /tmp/abp/log 2>&1)
        python - "$e" $ctr <<'PY'
import csv, glob, sys, collections, re
f = glob.glob('/tmp/abp/**/p_counter_collection.csv', recursive=True)
tot = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f[0])):
    if r['Counter_Name'] != sys.argv[2]: continue
    m = re.search(r'ntt_pass_[ab]', r['Kernel_Name'])
    if m: tot[m.group(0)][0] += 1; tot[m.group(0)][1] += float(r['Counter_Value'])
print('%-60s %-10s' % (sys.argv[1] or 'default', sys.argv[2]), '  '.join('%s: %.3f GB per proof (%d launches)' % (k, v[1] * 1024 / 4 / 1e9, v[0]) for k, v in sorted(tot.items())))   # 1 warm-up + 2 timed proofs + the instrumented one
PY
    done
done
