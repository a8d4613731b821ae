export BENCH_TRACE_CACHE=/tmp/dtc
for r in 1 2 3; do
 for lib in product nohints; do
  for cfg in "--log-n 22" "--log-n 24 --log-blowup 4 --queries 100"; do
   E="X_=1"; [ $lib = nohints ] && E="DISTAFF_HIP_LIB=gpurun_tmp_libs/nohints/distaff_amd/libdistaff_hip.so"
   env $E python bench.py $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-upload-leg --no-verify --allow-lib-override 2>/dev/null | grep "^{" | python -c "
import json,sys
b=json.loads(sys.stdin.read()); print('$lib', '$cfg'.split()[1], round(b['ms_per_step'],2), b['phase_ms']['lde'])"
  done
 done
done
