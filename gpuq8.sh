timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | head
bash tools/profile_round.sh r1 3 2>&1 | tail -3
cat gpurun_out/profiles_r1/r1_bench_under_rocprof.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline'], d['alu_roofline'])"
