#!/bin/bash
# round 4, GPU call 1 (through gpurun): the new -m gpu tests, then baseline numbers of the sizes / shapes the round works on.
export TMPDIR=/tmp BENCH_TRACE_CACHE=/tmp/dtc
O=gpurun_out/r4a; mkdir -p $O
df -h /tmp | tail -1 > $O/df.txt
( time timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "fresh_contexts or separate_processes or n_processes_on_one_device or error_line or config2_line or sampled" ) > $O/tests.log 2>&1
tail -3 $O/tests.log
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --log-n 22 --steps 3 --warmup 1 --no-cpu-baseline --no-upload-leg > $O/bench_22.json 2> $O/bench_22.err
timeout 600 python bench.py --log-n 24 --log-blowup 4 --queries 100 --steps 3 --warmup 1 --no-cpu-baseline --no-upload-leg > $O/bench_config5.json 2> $O/bench_config5.err
timeout 300 python bench.py --workload commit > $O/bench_config2.json 2> $O/bench_config2.err
for N in 2 8; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2950$N bench.py --gpus $N --steps 5 --warmup 2 > $O/bench_shared_$N.json 2> $O/bench_shared_$N.err
done
timeout 600 python tools/replicated_estimate.py > $O/replicated.txt 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4a/bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][0])
        print(f, d.get("error") or ("%.2f ms" % d["ms_per_step"]), d.get("phase_ms"))
    except Exception as e:
        print(f, "unreadable", e)
PY
cat $O/replicated.txt
