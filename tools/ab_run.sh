#!/bin/bash
# On the GPU box: the short bench for the base library (tools/ab_base.sh) and for the working tree under each given environment setting.
#   bash tools/ab_run.sh "" "DISTAFF_NTT_B8=1" ...
run() {
    env $1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-upload-leg --no-verify 2>&1 | grep "^{" | python -c "
import json, sys
b = json.loads(sys.stdin.read())
k = b['kernels']
print('%-44s %.3f ms  lde %.3f  pass_a %.3f  pass_b %.3f  air %.3f' % ('$2', b['ms_per_step'], b['phase_ms']['lde'], k.get('ntt_pass_a', {}).get('ms_per_step', 0), k.get('ntt_pass_b', {}).get('ms_per_step', 0), b['phase_ms']['constraint_eval']))"
}
[ -f gpurun_tmp_libs/base/distaff_amd/libdistaff_hip.so ] && run "DISTAFF_HIP_LIB=gpurun_tmp_libs/base/distaff_amd/libdistaff_hip.so" "base $(cat gpurun_tmp_libs/base/REV)"
for e in "$@"; do run "${e:-X_=1}" "tree ${e}"; done
[ -f gpurun_tmp_libs/base/distaff_amd/libdistaff_hip.so ] && run "DISTAFF_HIP_LIB=gpurun_tmp_libs/base/distaff_amd/libdistaff_hip.so" "base again"
