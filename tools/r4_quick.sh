#!/bin/bash
# quick A/B on one box: default bench (short) with env variants given as arguments "NAME:ENV=VAL,ENV2=VAL2" ...
export TMPDIR=/tmp BENCH_TRACE_CACHE=/tmp/dtc
O=gpurun_out/r4q; mkdir -p $O
python - <<'PY'
import bench, distaff_amd as D
bench.fibonacci_trace_cached(D, 20)
PY
for round in 1 2; do
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}; [ "$envs" = "$spec" ] && envs=""
  ( IFS=, read -ra kvs <<< "$envs"; for kv in "${kvs[@]}"; do [ -n "$kv" ] && export "$kv"; done
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-upload-leg --no-verify $EXTRA > $O/${name}_$round.json 2> $O/${name}_$round.err )
done; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4q/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][0])
        print(f.split("/")[-1], d.get("error") or ("%.3f ms  min %.3f" % (d["ms_per_step"], d["step_ms"]["min"])), {k: round(v, 2) for k, v in d["phase_ms"].items()})
    except Exception as e:
        print(f, "unreadable", e)
PY
