# SQ counters of a command, per kernel: gpurun -- 'bash tools/pmc.sh python tools/gpu_one_render.py cornell 32'
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_tmp; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $OUT/a -o r -- "$@" > $OUT/stdout.log 2> $OUT/err.log
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA -d $OUT/b -o r -- "$@" >> $OUT/stdout.log 2>> $OUT/err.log
if [ -n "$PMC_MEM" ]; then
rocprofv3 --pmc FETCH_SIZE -d $OUT/c -o r -- "$@" >> $OUT/stdout.log 2>> $OUT/err.log
rocprofv3 --pmc WRITE_SIZE -d $OUT/d -o r -- "$@" >> $OUT/stdout.log 2>> $OUT/err.log
fi
tail -3 $OUT/stdout.log
python tools/rocpd_summary.py pmc $(find $OUT -name "*.db") | grep -E "${PMC_GREP:-k_render|k_first|k_continue|k_fold|k_primary}" | sort
find $OUT -name "*.db" -delete
