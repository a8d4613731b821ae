timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fibonacci_all or blowup or config2 or lds" 2>&1 | grep -E "passed|failed|Error|error" | head
python /tmp/t20.py 2>/dev/null || true
python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step']); print(d['phase_ms']); print({k:v['ms_per_step'] for k,v in list(d['kernels'].items())[:9]})"
