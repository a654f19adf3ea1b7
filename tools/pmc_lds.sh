# LDS counters of a command, per kernel
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_tmp; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_ADDR_CONFLICT -d $OUT/a -o r -- "$@" > $OUT/stdout.log 2> $OUT/err.log
tail -2 $OUT/stdout.log; tail -3 $OUT/err.log | cut -c1-200
python tools/rocpd_summary.py pmc $(find $OUT -name "*.db") | grep -E "${PMC_GREP:-k_render|k_continue|k_fold|k_primary|k_intersect}" | sort
find $OUT -name "*.db" -delete
