#!/bin/bash
# On the GPU box (through gpurun): the instruction-cache evidence of one lease.
#   1. tools/icache_probe: time per instruction of straight-line kernels of 16 ... 256 KiB, plain and under the SQC_ICACHE_* counters
#   2. the constraint kernels of the bench through the product library and every variant library under gpurun_tmp_libs/
#   3. the SQC_ICACHE_* / SQ_IFETCH counters of the bench's kernels (product library)
# Results: gpurun_out/icache_<tag>/ (probe.json, probe_pmc.csv, air_ab.txt, bench_pmc.csv); summarise with tools/icache_summary.py
TAG=${1:-lease}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/icache_$TAG
rm -rf $OUT; mkdir -p $OUT
CTRS="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"
(rocm-smi --showclocks --showmemorypartition --showcomputepartition 2>&1 | grep -v "^$" | head -40) > $OUT/rocm_smi.txt
$R/tools/icache_probe/_build/icache_probe --json > $OUT/probe.json 2> $OUT/probe.err
$R/tools/icache_probe/_build/icache_probe | tee $OUT/probe.txt
cd /tmp
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/probe_pmc -o p -- $R/tools/icache_probe/_build/icache_probe > $OUT/probe_pmc.log 2>&1
cd $R
run() {
    env $1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-upload-leg --no-verify 2>&1 | grep "^{" | python -c "
import json, sys
b = json.loads(sys.stdin.read())
k = b['kernels']
air = {n: v['ms_per_step'] for n, v in k.items() if n.startswith('air_kernel')}
print('%-28s %.3f ms  constraint_eval %.3f  ' % ('$2', b['ms_per_step'], b['phase_ms']['constraint_eval']) + '  '.join('%s %.3f' % (n[10:], v) for n, v in sorted(air.items())))"
}
{
run "X_=1" "product"
for d in gpurun_tmp_libs/*/; do
    n=$(basename $d)
    [ -f $d/distaff_amd/libdistaff_hip.so ] && run "DISTAFF_HIP_LIB=$d/distaff_amd/libdistaff_hip.so" "$n"
done
run "X_=1" "product again"
} | tee $OUT/air_ab.txt
cd /tmp
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/bench_pmc -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-upload-leg --no-verify > $OUT/bench_pmc.log 2>&1
cd $R
python tools/icache_summary.py $OUT | tee $OUT/summary.txt
