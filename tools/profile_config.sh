#!/bin/bash
# rocprofv3 summary of bench.py on ANOTHER configuration than the default one (BASELINE configs 2, 4, 5): kernel-trace/stats pass + FETCH_SIZE,
# WRITE_SIZE and one SQ pass, each its own run (counters are never combined with other trace domains), summarised like tools/profile_round.sh.
#   usage: bash tools/profile_config.sh <tag> "<bench.py arguments>"        -> gpurun_out/profiles_<tag>/<tag>_*
TAG=$1; ARGS=$2
export TMPDIR=/tmp BENCH_TRACE_CACHE=${BENCH_TRACE_CACHE:-/tmp/dtc}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT $R/gpurun_out/profiles_$TAG
python $R/bench.py $ARGS 2> $OUT/bench.err | grep "^{" > $R/gpurun_out/profiles_$TAG/${TAG}_bench.json      # the line of its own, no profiler (CPU leg included)
cd /tmp
CMD="python $R/bench.py $ARGS --steps 3 --warmup 1 --no-cpu-baseline --no-upload-leg --no-verify"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o p -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o p -- $CMD > $OUT/write.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/grbm -o p -- $CMD > $OUT/grbm.log 2>&1      # busy cycles per launch: the clock the kernels ran at
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/sq1 -o p -- $CMD > $OUT/sq1.log 2>&1
cd $R
grep -h "^{" $OUT/stats.log > $OUT/bench_under_rocprof.json
# rocprofv3 nests its output under the host name: flatten what summarize_profile.py expects
for sub in stats fetch write sq1 grbm; do f=$(find $OUT/$sub -name "p_kernel_stats.csv" -o -name "p_counter_collection.csv" -o -name "p_kernel_trace.csv" | head -5); for x in $f; do cp $x $OUT/$sub/ 2>/dev/null; done; done
python tools/summarize_profile.py $OUT $R/gpurun_out/profiles_$TAG $TAG
python - <<PY
import json, os, sys
sys.path.insert(0, "$R")
import bench
info = {}
try:
    info = json.load(open(os.path.join("$R", "distaff_amd", "_build_info.json")))
except Exception:
    pass
json.dump({"tag": "$TAG", "csrc_sha16": bench.csrc_digest(), "git_head": info.get("git_head"), "command": "$CMD"},
          open(os.path.join("$R", "gpurun_out", "profiles_$TAG", "${TAG}_meta.json"), "w"), indent=1)
PY
rm -rf $OUT
ls -la $R/gpurun_out/profiles_$TAG
