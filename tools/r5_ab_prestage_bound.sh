export BENCH_TRACE_CACHE=/tmp/dtc
for r in 1 2 3; do
 for lib in product presharedbound; do
   E="X_=1"; [ $lib != product ] && E="DISTAFF_HIP_LIB=gpurun_tmp_libs/$lib/distaff_amd/libdistaff_hip.so"
   env $E python bench.py --log-n 22 --steps 3 --warmup 1 --no-cpu-baseline --no-upload-leg --no-verify --allow-lib-override 2>/dev/null | grep "^{" | python -c "
import json,sys
b=json.loads(sys.stdin.read()); k=b['kernels']; print('$lib', round(b['ms_per_step'],2), b['phase_ms']['lde'], k['ntt_pass_a']['ms_per_step'], k['ntt_pass_b']['ms_per_step'])"
 done
done
