set -x
export TMPDIR=/tmp
R=$PWD
python bench.py --steps 5 --warmup 2 > gpurun_out/bench_r1_a.json 2> gpurun_out/bench_r1_a.err
tail -c 3000 gpurun_out/bench_r1_a.json
cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r1_a -o prover -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_r1_a.log 2>&1
cd $R
find gpurun_out/prof_r1_a -name "*stats*" | head
f=$(find gpurun_out/prof_r1_a -name "*kernel_stats.csv" | head -1); head -30 "$f"
