# Round-2 profile of the default bench line (config 2 + the HBM-resident 3840x2160 Poisson block): kernel trace, then separate PMC passes.
#   gpurun --timeout 1500 -- 'TAG=r02a bash tools/prof_r02.sh'    ->  gpurun_out/prof_<TAG>/{kernel_stats.csv,pmc.csv,counters.json,...}
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r02a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-}"
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/kt -o r1 -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o r1 -- $B > /dev/null 2> $OUT/pmc_fetch.err
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o r1 -- $B > /dev/null 2> $OUT/pmc_write.err
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_sq1 -o r1 -- $B > /dev/null 2> $OUT/pmc_sq1.err
timeout 400 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA -d $OUT/pmc_sq2 -o r1 -- $B > /dev/null 2> $OUT/pmc_sq2.err
python tools/rocpd_summary.py stats $(find $OUT/kt -name "*.db" | head -1) > $OUT/kernel_stats.csv
python tools/rocpd_summary.py pmc $(find $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq1 $OUT/pmc_sq2 -name "*.db") > $OUT/pmc.csv
python tools/profile_json.py $TAG $OUT/counters.json $(find $OUT/kt -name "*.db" | head -1) $(find $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq1 $OUT/pmc_sq2 -name "*.db")
find $OUT -name "*.db" -size +20M -delete
head -12 $OUT/kernel_stats.csv; grep -E "k_render|k_first|k_continue|kp_cg|kf_xp_Ax" $OUT/pmc.csv | head -40; tail -1 $OUT/bench_under_rocprof.json | cut -c1-1500
tail -2 $OUT/*.err | cut -c1-300
