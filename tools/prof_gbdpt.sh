# PMC profile of the G-BDPT sampler at config 5's scene and size (few samples per pixel): the counters bench.py --config 5 prices per sample.
#   gpurun --timeout 1500 -- 'TAG=r04a bash tools/prof_gbdpt.sh'   ->  gpurun_out/prof_<TAG>_gbdpt/{kernel_stats.csv,pmc.csv,counters_gbdpt.json}
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r04a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_gbdpt
rm -rf $OUT; mkdir -p $OUT
B="python tools/gpu_gbdpt_perf.py ${GBDPT_SPP:-2} ${GBDPT_SCENE:-veach_specular}"      # (config 5's scene since round 4: with the glass egg, the mirror and the polished copper)
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/kt -o r1 -- $B > $OUT/stdout.log 2> $OUT/kt.err
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o r1 -- $B > /dev/null 2> $OUT/pmc_fetch.err
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o r1 -- $B > /dev/null 2> $OUT/pmc_write.err
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_sq1 -o r1 -- $B > /dev/null 2> $OUT/pmc_sq1.err
python tools/rocpd_summary.py stats $(find $OUT/kt -name "*.db" | head -1) > $OUT/kernel_stats.csv
python tools/rocpd_summary.py pmc $(find $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq1 -name "*.db") > $OUT/pmc.csv
python tools/profile_json.py ${TAG}_gbdpt $OUT/counters_gbdpt.json $(find $OUT/kt -name "*.db" | head -1) $(find $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq1 -name "*.db")
find $OUT -name "*.db" -delete
head -12 $OUT/kernel_stats.csv; tail -3 $OUT/stdout.log; tail -2 $OUT/*.err | cut -c1-300
