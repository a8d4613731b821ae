#!/bin/bash
# On the GPU box: the constraint-kernel parity tests, then tools/icache_round.sh (probe + AIR timings of the product and of every library
# under gpurun_tmp_libs/ + instruction-cache counters).   bash tools/gpu_air_round.sh <tag>
TAG=${1:-lease}
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fibonacci_all_phases or general_constraint or other_program_shapes or stack_depth_5 or deep_stacks or invalid_trace or boundary_constraints or device_field or config3 or golden" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -6
bash tools/icache_round.sh $TAG 2>&1 | grep -v "^code\|probe<" | tail -40
