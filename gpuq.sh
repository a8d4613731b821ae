python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_q.json 2> gpurun_out/bench_q.err || tail -5 gpurun_out/bench_q.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_q.json'))
print("ms/step %.2f cells/s %.3e" % (d['ms_per_step'], d['value']))
print(d['phase_ms'])
for k,v in list(d['kernels'].items())[:8]: print(k, v)
print("mulmod peak %.1f G/s" % (d['alu_roofline']['peak_measured']/1e9))
import distaff_amd as D
ctx = D.Context(10, 20, 1, 0)
for port in (False, True):
    ms = ctx.bench_mulmod(1<<22, 256, portable=port); print("portable" if port else "gfx950", "%.1f G mulmod/s" % ((1<<22)*256*4/ms/1e6))
PY
