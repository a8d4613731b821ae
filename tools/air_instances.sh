timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "general_constraint or shapes" 2>&1 | grep -E "passed|failed|Error|error|assert" | head -5
for f in sd4 small generic; do if [ $f = sd4 ]; then unset DISTAFF_AIR; else export DISTAFF_AIR=$f; fi; python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); print(os.environ.get('DISTAFF_AIR','sd4'), round(d['ms_per_step'],2), d['phase_ms']['constraint_eval'])"; done
