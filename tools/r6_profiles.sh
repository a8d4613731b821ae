#!/bin/bash
# round 6, on the GPU box (through gpurun): the rocprofv3 summaries of the final build for every configuration (stats + FETCH_SIZE + WRITE_SIZE +
# GRBM_GUI_ACTIVE [the clock the kernels ran at] + SQ passes), then -- with those summaries copied into profiles/ ON THE BOX, so that bench.py finds
# counters whose stamp matches the sources -- the bench line of every configuration, the one-rank RCCL line, the N-process functional runs, and
# the CPU leg at the headline size.  Everything lands under gpurun_out/profiles_r6*/ and gpurun_out/r6_lines/.
export TMPDIR=/tmp BENCH_TRACE_CACHE=/tmp/dtc
mkdir -p gpurun_out
bash tools/profile_round.sh r6 3 > gpurun_out/r6_profile_round.log 2>&1
bash tools/profile_config.sh r6_config2 "--workload commit" > gpurun_out/r6_config2.log 2>&1
bash tools/profile_config.sh r6_config4 "--log-n 22" > gpurun_out/r6_config4.log 2>&1
bash tools/profile_config.sh r6_config5 "--log-n 24 --log-blowup 4 --queries 100" > gpurun_out/r6_config5.log 2>&1
for d in gpurun_out/profiles_r6 gpurun_out/profiles_r6_config2 gpurun_out/profiles_r6_config4 gpurun_out/profiles_r6_config5; do cp $d/*_pmc_per_kernel.csv $d/*_meta.json profiles/ 2>/dev/null; done
O=gpurun_out/r6_lines; mkdir -p $O
timeout 400 python bench.py > $O/r6_bench_default.json 2> $O/default.err
timeout 300 python bench.py --workload commit > $O/r6_config2_bench.json 2> $O/c2.err
timeout 600 python bench.py --log-n 22 > $O/r6_config4_bench.json 2> $O/c4.err
timeout 900 python bench.py --log-n 24 --log-blowup 4 --queries 100 > $O/r6_config5_bench.json 2> $O/c5.err
timeout 400 python bench.py --force-sharded --no-cpu-baseline > $O/r6_bench_force_sharded_rccl_1rank.json 2> $O/fs.err
for N in 2 8; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2950$N bench.py --gpus $N --steps 5 --warmup 2 > $O/r6_bench_shared_device_$N.json 2> $O/shared_$N.err
done
timeout 900 python bench.py --cpu-log-n 20 --steps 5 --warmup 2 --no-upload-leg > $O/r6_bench_cpu_2_20.json 2> $O/cpu20.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6_lines/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][0]); r = d.get("roofline") or {}
        vi = (d.get("alu_roofline") or {}).get("valu_issue") or {}
        print(f, d.get("error") or ("%.2f ms" % d["ms_per_step"]), "traffic x", r.get("traffic_over_algorithmic"), "valu", r.get("valu_issue_frac"), "sclk", r.get("sclk_MHz"), "at clock", r.get("valu_issue_frac_at_measured_clock"),
              "proof", vi.get("proof_frac"), vi.get("proof_frac_at_measured_clock"), "box clock", (d.get("box") or {}).get("sclk_under_load_MHz"), (d.get("comm") or {}).get("transport"), d.get("exchange_ms_rank0"))
    except Exception as e:
        print(f, "unreadable", e)
PY
