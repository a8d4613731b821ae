#!/bin/bash
# On the GPU box: a pytest selection, then the bench without its CPU legs; prints total, phases and the kernels above 0.05 ms.
#   bash tools/gpu_bench_short.sh [pytest -k expression]
[ -n "$1" ] && bash tools/gpu_check.sh "$1"
mkdir -p gpurun_out
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-upload-leg 2>&1 | grep "^{" > gpurun_out/b.json
python - <<PY
import json
b = json.loads(open("gpurun_out/b.json").read())
print("ms", round(b["ms_per_step"], 3), b["phase_ms"])
for k, v in b["kernels"].items():
    if v["ms_per_step"] >= 0.05: print("  %-34s %3d  %.3f ms  %6.0f GB/s" % (k[:34], v["launches"], v["ms_per_step"], v["GBps"]))
PY
