timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sharded" 2>&1 | grep -E "passed|failed|Error|error|assert" | head
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --force-sharded 2>/dev/null | grep '^{"metric' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['parallelism']); print(d['shard_stage_ms_rank0'])"
