#!/bin/bash
# On the GPU box (through gpurun): the GPU suite with a readable tail (RCCL prints its banner at exit), then the default bench line.
#   bash tools/gpu_check.sh [pytest -k expression]
python -m pytest tests -m gpu -x -q ${1:+-k "$1"} 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -6
