#!/bin/bash
# On the GPU box: the straight-line-code probe of the library (dst_bench_code, a few seconds without torch); when this lease shows the slow
# instruction delivery (176 KiB of code > 1.02 per instruction relative to 16 KiB; healthy: 0.90 - 0.93), the whole instruction-cache
# round with counters is taken at once (tools/icache_round.sh slow_<tag>).      bash tools/slow_box_hunt.sh <tag>
TAG=${1:-hunt}
mkdir -p gpurun_out/hunt
python - "$TAG" <<'PY' | tee gpurun_out/hunt/$TAG.txt
import sys
sys.path.insert(0, ".")
import distaff_amd as D
ctx = D.Context(10, 20, 1, 0)
t16, t176, t177 = ctx.bench_code(16), ctx.bench_code(176), ctx.bench_code(177)
mad = ctx.bench_mad(1 << 21, 512)
ratio = (t176 / 176.0) / (t16 / 16.0)
print("lease %s: code probe 16 KiB %.4f ms, 176 KiB %.4f ms -> %.3f per instruction; convoy form %.4f ms -> %.3f; mad calibration %.3f ms" % (sys.argv[1], t16, t176, ratio, t177, (t177 / 176.0) / (t16 / 16.0), mad))
print("SLOW" if ratio > 1.02 else "healthy")
ctx.close()
PY
if grep -q "^SLOW" gpurun_out/hunt/$TAG.txt; then bash tools/icache_round.sh slow_$TAG 2>&1 | tail -60; fi
