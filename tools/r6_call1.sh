#!/bin/bash
# round 6, first measurement call: containment tests, sharded phase times, laboratory bounds of the first pass, calibrations, stamps
export TMPDIR=/tmp BENCH_TRACE_CACHE=/tmp/dtc
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "peer_that_never_arrives or stalled_collective or stalled_peer_process or behind_the_c_abi or rank_without_a_trace or thread_rank_transport or separate_processes" 2>&1 | tail -15 ) > gpurun_out/r6_tests_containment.log 2>&1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-upload-leg > gpurun_out/r6_bench_single.json 2> gpurun_out/r6_bench_single.err
python bench.py --force-sharded --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r6_bench_force_sharded.json 2> gpurun_out/r6_bench_force_sharded.err
python - <<'PY' > gpurun_out/r6_phase_compare.txt
import json
a = json.loads(open("gpurun_out/r6_bench_single.json").read().strip().splitlines()[-1])
b = json.loads(open("gpurun_out/r6_bench_force_sharded.json").read().strip().splitlines()[-1])
print("single context  %.3f ms" % a["ms_per_step"], a["phase_ms"])
print("force-sharded   %.3f ms" % b["ms_per_step"], b["phase_ms"])
print("exchange", b.get("exchange_ms_rank0"), "stage", b.get("shard_stage_ms_rank0"), "comm", b.get("comm"))
PY
cat gpurun_out/r6_phase_compare.txt
bash tools/r6_lde_lab.sh 3 "20 22" > gpurun_out/r6_lde_lab_summary.txt 2>&1
cat gpurun_out/r6_lde_lab_summary.txt
bash tools/r6_calibrate.sh > gpurun_out/r6_calibrate.log 2>&1
cat gpurun_out/r6_cal/issue_slots.txt gpurun_out/r6_cal/fetch_factor.txt
DISTAFF_HIP_LIB=gpurun_tmp_libs/stamps/distaff_amd/libdistaff_hip.so python tools/r6_pass_stamps.py run 20 gpurun_out/r6_stamps_20.json > gpurun_out/r6_stamps.log 2>&1
python tools/r6_pass_stamps.py table gpurun_out/r6_stamps_20.json | head -80
cat gpurun_out/r6_tests_containment.log
