#!/bin/bash
# round 5, on the GPU box (through gpurun): counter summaries + bench lines of BASELINE configs 2, 4, 5 on the final build, the default line
# again (now that profiles/r5_pmc_per_kernel.csv of this build is in the tree it carries traffic / VALU issue), the N-process path on one device.
export TMPDIR=/tmp BENCH_TRACE_CACHE=/tmp/dtc
mkdir -p gpurun_out
bash tools/profile_config.sh r5_config2 "--workload commit" > gpurun_out/r5_config2.log 2>&1
bash tools/profile_config.sh r5_config4 "--log-n 22" > gpurun_out/r5_config4.log 2>&1
bash tools/profile_config.sh r5_config5 "--log-n 24 --log-blowup 4 --queries 100" > gpurun_out/r5_config5.log 2>&1
timeout 400 python bench.py > gpurun_out/r5_bench_default_with_counters.json 2> gpurun_out/r5_bench_default_with_counters.err
for N in 2 8; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2950$N bench.py --gpus $N --steps 5 --warmup 2 > gpurun_out/r5_bench_shared_device_$N.json 2> gpurun_out/r5_bench_shared_device_$N.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/profiles_r5_config*/*_bench.json")) + sorted(glob.glob("gpurun_out/r5_bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][0])
        r = d.get("roofline") or {}
        print(f, d.get("error") or ("%.2f ms" % d["ms_per_step"]), "traffic", r.get("traffic"), "valu", r.get("valu_issue_frac"), (d.get("comm") or {}).get("transport"))
    except Exception as e:
        print(f, "unreadable", e)
PY
