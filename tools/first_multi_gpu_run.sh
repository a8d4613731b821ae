#!/bin/bash
# For the first node with more than one MI355X (no round of this build has had one): the scaling curve and the three comparisons DESIGN.md section 6
# asks for, each a bench line under gpurun_out/multi/.      bash tools/first_multi_gpu_run.sh [max ranks, default = number of GPUs]
#   1. bench.py --gpus 1, 2, 4, 8 as the driver launches it (config 3): the line's `comm` must say transport rccl and ranks_per_rank = [N] * N
#   2. the same at N = max with every collective on the main stream (DISTAFF_SHARD_NO_OVERLAP=1): what the second stream buys
#   3. the same with the collectives carried by torch.distributed instead of the library's RCCL binding (DISTAFF_SHARD_TRANSPORT=callbacks)
#   4. the all-gather-only tree form of BASELINE's north_star (DISTAFF_SHARD_TREE_GATHER=1)
#   5. configs 4 and 5 at N = max
# Compare `phase_ms` (events on the prover's stream), `exchange_ms_rank0` (enqueue -> completion of every collective kind, i.e. incl. the wait for
# the slowest peer) and `shard_stage_ms_rank0` with the per-rank models of DESIGN.md section 6.  No rank can hang: every host wait behind a collective
# is bounded (DISTAFF_COMM_TIMEOUT_S, 60 s by default; on expiry the rank aborts its communicator and bench.py prints ONE line with the error, which
# names the collective the rank was stuck behind -- dst_comm_last_error).
set -u
export TMPDIR=/tmp BENCH_TRACE_CACHE=${BENCH_TRACE_CACHE:-/tmp/dtc}
R=$(cd "$(dirname "$0")/.." && pwd); cd "$R"
O=gpurun_out/multi; mkdir -p $O
NDEV=$(python -c "import torch; print(torch.cuda.device_count())")
MAX=${1:-$NDEV}
run() {   # name, ranks, extra bench arguments, environment assignments ...
    local name=$1 n=$2 args=$3; shift 3
    if [ "$n" = 1 ]; then env "$@" X_=1 timeout 900 python bench.py --gpus 1 $args > $O/$name.json 2> $O/$name.err
    else env "$@" X_=1 timeout 1800 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n $args > $O/$name.json 2> $O/$name.err; fi
    python - "$O/$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d.get("error") or "%.2f ms  %.3e cells/s" % (d["ms_per_step"], d["value"]), (d.get("comm") or {}).get("transport"), (d.get("comm") or {}).get("ranks_per_rank"), d.get("shard_stage_ms_rank0"))
    print("    phases", d.get("phase_ms"), "\n    exchanges", d.get("exchange_ms_rank0"))
except Exception as e:
    print(sys.argv[1], "no line:", e)
PY
}
for n in 1 2 4 8; do [ $n -le $MAX ] && run scale_$n $n ""; done
if [ $MAX -gt 1 ]; then
    run no_overlap_$MAX $MAX "" DISTAFF_SHARD_NO_OVERLAP=1
    run callbacks_$MAX $MAX "" DISTAFF_SHARD_TRANSPORT=callbacks
    run tree_gather_$MAX $MAX "" DISTAFF_SHARD_TREE_GATHER=1
    run config4_$MAX $MAX "--log-n 22 --steps 5 --warmup 2"
    run config5_$MAX $MAX "--log-n 24 --log-blowup 4 --queries 100 --steps 3 --warmup 1"
fi
