# kernel trace of a command: gpurun -- 'bash tools/kt.sh python tools/gpu_one_render.py cornell 32'
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/kt_tmp; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o r -- "$@" > $OUT/stdout.log 2> $OUT/err.log
tail -4 $OUT/stdout.log
python tools/rocpd_summary.py stats $(find $OUT -name "*.db" | head -1) | head -${KT_LINES:-12}
find $OUT -name "*.db" -delete
