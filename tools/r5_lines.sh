#!/bin/bash
# round 5, last GPU call: the bench lines of every configuration with the stamped r5 counter summaries in the tree (traffic / VALU issue in the line)
export TMPDIR=/tmp BENCH_TRACE_CACHE=/tmp/dtc
O=gpurun_out/r5_lines; mkdir -p $O
timeout 400 python bench.py > $O/r5_bench_default.json 2> $O/default.err
timeout 300 python bench.py --workload commit > $O/r5_config2_bench.json 2> $O/c2.err
timeout 600 python bench.py --log-n 22 > $O/r5_config4_bench.json 2> $O/c4.err
timeout 900 python bench.py --log-n 24 --log-blowup 4 --queries 100 > $O/r5_config5_bench.json 2> $O/c5.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5_lines/*.json")):
    d = json.loads([l for l in open(f) if l.startswith("{")][0]); r = d.get("roofline") or {}
    print(f, d.get("error") or ("%.2f ms" % d["ms_per_step"]), "traffic", r.get("traffic"), "x", r.get("traffic_over_algorithmic"), "valu", r.get("valu_issue_frac"), "frac", r.get("frac"))
PY
