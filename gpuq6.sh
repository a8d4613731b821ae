for v in 0 1; do
for ln in 18 22; do
if [ $v = 1 ]; then export DISTAFF_NTT_LDS=1; else unset DISTAFF_NTT_LDS; fi
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --log-n $ln 2>&1 | grep metric | python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); print('lds' if os.environ.get('DISTAFF_NTT_LDS') else 'reg', d['config']['trace_steps'], round(d['ms_per_step'],2), d['phase_ms']['lde'], {k:v['ms_per_step'] for k,v in list(d['kernels'].items())[:4]})"
done; done
