# every kernel dispatch of a command (name prefix, grid, duration): gpurun -- 'timeout 300 bash tools/kt_list.sh python tools/gpu_gbdpt_perf.py 4 veach'
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/kt_tmp; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT -o r -- "$@" > $OUT/stdout.log 2> $OUT/err.log
tail -3 $OUT/stdout.log
python - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob("$OUT/*.db")[0])
for name, grid, dur in db.execute("select name, grid_x, duration/1e3 from kernels order by start"):
    if "rocclr" in name: continue
    print("%-40s grid %10d  %10.1f us" % (name[name.find("k_"):][:40] if "k_" in name else name[:40], grid, dur))
PY
find $OUT -name "*.db" -delete
