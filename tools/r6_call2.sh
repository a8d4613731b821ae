#!/bin/bash
# round 6, second measurement call: issue slots with pinned residency, data dependence of the passes, clocks under load, lab bounds again
export TMPDIR=/tmp BENCH_TRACE_CACHE=/tmp/dtc
mkdir -p gpurun_out/r6_cal2
tools/felab/_build/issuelab issue > gpurun_out/r6_cal2/issue_slots.txt 2>&1
cat gpurun_out/r6_cal2/issue_slots.txt
# clocks while the prover runs: rocm-smi samples beside a 300-proof run
( python bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-upload-leg --no-verify > gpurun_out/r6_cal2/bench_300.json 2>/dev/null & 
  sleep 14; for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (edge|junction|hotspot)" ; echo --; sleep 1.5; done; wait ) > gpurun_out/r6_cal2/clocks_under_load.txt 2>&1
cat gpurun_out/r6_cal2/clocks_under_load.txt | head -60
python -c "
import json; d=json.loads(open('gpurun_out/r6_cal2/bench_300.json').read().strip().splitlines()[-1]); print('300 proofs:', d['ms_per_step'], d['step_ms']['min'], d['step_ms']['median'], d['step_ms']['max'])"
# the same instructions on constant columns
: > gpurun_out/r6_cal2/data_dependence.txt
for r in 1 2 3; do for d in random zeros ones; do python tools/r6_lde_lab.py product_$d 20 3 $d >> gpurun_out/r6_cal2/data_dependence.txt 2>/dev/null; done; done
cat gpurun_out/r6_cal2/data_dependence.txt
bash tools/r6_lde_lab.sh 3 "20 22" > gpurun_out/r6_lde_lab_summary2.txt 2>&1
cat gpurun_out/r6_lde_lab_summary2.txt
DISTAFF_HIP_LIB=gpurun_tmp_libs/stamps/distaff_amd/libdistaff_hip.so python tools/r6_pass_stamps.py run 20 gpurun_out/r6_stamps_20.json > gpurun_out/r6_stamps.log 2>&1
python tools/r6_pass_stamps.py table gpurun_out/r6_stamps_20.json | grep -E "tile|###" 
cd /tmp; rocprofv3 -L 2>/dev/null | grep -i -E "TCC_EA0_RDREQ|TCC_EA0_WRREQ|TCC_BUBBLE|TCC_EA0_RD_UNCACHED" | head -30 > $GRAFT_REPO_ROOT/gpurun_out/r6_cal2/counters.txt; cat $GRAFT_REPO_ROOT/gpurun_out/r6_cal2/counters.txt | cut -c1-200
