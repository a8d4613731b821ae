#!/bin/bash
# timing ablations of the LDE passes (results are wrong by construction): DISTAFF_NTT_DEBUG bits 2 = no four-step twiddle, 4 = no stage
# multiplications, 8 = no additions / subtractions
for d in ${ABL:-0 2 4 6 12 14}; do
  DISTAFF_NTT_DEBUG=$d python bench.py --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('debug=$d', 'lde', d['phase_ms']['lde'], 'pass_a', round(k['ntt_pass_a']['ms_per_step'],2), 'pass_b', round(k['ntt_pass_b']['ms_per_step'],2))"
done
