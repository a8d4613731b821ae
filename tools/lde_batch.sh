#!/bin/bash
# LDE time against the launch granularity (registers x cosets per pair of passes): does a staging buffer that fits the 256 MiB
# Infinity Cache save HBM time?
for b in "4,31" "2,31" "1,31" "1,16" "1,8" "2,8" "4,8" "1,4" "4,4"; do
  DISTAFF_LDE_BATCH=$b python bench.py --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('batch=$b', 'lde', d['phase_ms']['lde'], 'pass_a', round(k['ntt_pass_a']['ms_per_step'],2), 'pass_b', round(k['ntt_pass_b']['ms_per_step'],2), 'total', round(d['ms_per_step'],2))"
done
