#!/bin/bash
# quick timing of other sizes with env variants:  bash tools/r4_sizes.sh "<bench args>" NAME:ENV=VAL ...
export TMPDIR=/tmp BENCH_TRACE_CACHE=/tmp/dtc
O=gpurun_out/r4s; mkdir -p $O
ARGS="$1"; shift
tag=$(echo "$ARGS" | tr -c 'a-zA-Z0-9' '_')
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}; [ "$envs" = "$spec" ] && envs=""
  ( IFS=, read -ra kvs <<< "$envs"; for kv in "${kvs[@]}"; do [ -n "$kv" ] && export "$kv"; done
    python bench.py $ARGS --steps 4 --warmup 2 --no-cpu-baseline --no-upload-leg > $O/${name}${tag}.json 2> $O/${name}${tag}.err )
  python - <<PY
import json
f="$O/${name}${tag}.json"
try:
    d = json.loads([l for l in open(f) if l.startswith("{")][0])
    print("${name} $ARGS", d.get("error") or ("%.2f ms" % d["ms_per_step"]), {k: round(v, 2) for k, v in (d.get("phase_ms") or {}).items()}, d.get("proof_verified", "")[:12])
except Exception as e:
    print(f, "unreadable", e, open("$O/${name}${tag}.err").read()[-600:])
PY
done
