#!/bin/bash
# Quick GPU round trip used during development (run through gpurun from the repo root): the parity tests that finish in seconds,
# then three timed proofs with the per-phase and per-kernel breakdown.
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not every_tile" 2>&1 | grep -E "passed|failed|Error|error" | head
python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['alu_roofline']['peak_measured']); print(d['phase_ms']); print({k:v['ms_per_step'] for k,v in list(d['kernels'].items())[:9]})"
