set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG:-r01c}
rm -rf $OUT; mkdir -p $OUT
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o r1 -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o r1 -- $B > /dev/null 2> $OUT/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o r1 -- $B > /dev/null 2> $OUT/pmc_write.err
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY -d $OUT/pmc_sq1 -o r1 -- $B > /dev/null 2> $OUT/pmc_sq1.err
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA -d $OUT/pmc_sq2 -o r1 -- $B > /dev/null 2> $OUT/pmc_sq2.err
find $OUT -name "*.db" | head
for d in kt; do python tools/rocpd_summary.py stats $(find $OUT/$d -name "*.db" | head -1) > $OUT/kernel_stats.csv; done
python tools/rocpd_summary.py pmc $(find $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq1 $OUT/pmc_sq2 -name "*.db") > $OUT/pmc.csv
find $OUT -name "*.db" -size +20M -delete
head -30 $OUT/kernel_stats.csv; grep -E "k_render|kp_cg" $OUT/pmc.csv; tail -2 $OUT/bench_under_rocprof.json | cut -c1-600
tail -3 $OUT/*.err | cut -c1-300
