#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel-trace/stats pass + separate PMC passes over the SAME bench command, then
# summarises them into profiles/<tag>_* (copied back through gpurun_out/).   usage: bash tools/profile_round.sh <tag> [steps]
TAG=${1:-r1}; STEPS=${2:-3}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
CMD="python $R/bench.py --steps $STEPS --warmup 1 --no-cpu-baseline --no-upload-leg"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o p -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o p -- $CMD > $OUT/write.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/grbm -o p -- $CMD > $OUT/grbm.log 2>&1      # busy cycles per launch: the clock the kernels ran at
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/sq1 -o p -- $CMD > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/sq2 -o p -- $CMD > $OUT/sq2.log 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH --kernel-trace --output-format csv -d $OUT/sq3 -o p -- $CMD > $OUT/sq3.log 2>&1
cd $R
grep -h metric $OUT/stats.log > $OUT/bench_under_rocprof.json
python tools/summarize_profile.py $OUT $R/gpurun_out/profiles_$TAG $TAG
# stamp: the kernel sources these numbers were measured on (bench.py refuses a PMC summary whose stamp differs from the sources it runs)
python - <<PY
import json, os, sys
sys.path.insert(0, "$R")
import bench
info = {}
try:
    info = json.load(open(os.path.join("$R", "distaff_amd", "_build_info.json")))
except Exception:
    pass
json.dump({"tag": "$TAG", "csrc_sha16": bench.csrc_digest(), "git_head": info.get("git_head"), "command": "$CMD",
           "note": "git_head is the commit the shared library was built at (the GPU box has no .git); csrc_sha16 is the digest of distaff_amd/csrc at the time of the run"},
          open(os.path.join("$R", "gpurun_out", "profiles_$TAG", "${TAG}_meta.json"), "w"), indent=1)
PY
python bench.py 2>&1 | grep "^{" > $R/gpurun_out/profiles_$TAG/${TAG}_bench_default.json
ls -la $R/gpurun_out/profiles_$TAG
