export TMPDIR=/tmp
R=$PWD
cd /tmp
CMD="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o p -- $CMD > $R/gpurun_out/prof_stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/gpurun_out/prof_pmc1 -o p -- $CMD > $R/gpurun_out/prof_pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/prof_pmc2 -o p -- $CMD > $R/gpurun_out/prof_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_pmc3 -o p -- $CMD > $R/gpurun_out/prof_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_pmc4 -o p -- $CMD > $R/gpurun_out/prof_pmc4.log 2>&1
cd $R
find gpurun_out/prof_stats gpurun_out/prof_pmc1 -type f | head -20
