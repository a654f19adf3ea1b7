#!/bin/bash
# PMC passes over the atrium frame (HBM-resident BVH, 4-wave build of k_render): separate rocprofv3 runs per counter group
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_atrium
rm -rf $OUT; mkdir -p $OUT
B="python $R/tools/gpu_atrium_frame.py"
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/kt -o r1 -- $B > $OUT/run.log 2> $OUT/kt.err
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o r1 -- $B > /dev/null 2> $OUT/pmc_fetch.err
timeout 200 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o r1 -- $B > /dev/null 2> $OUT/pmc_write.err
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_sq1 -o r1 -- $B > /dev/null 2> $OUT/pmc_sq1.err
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA -d $OUT/pmc_sq2 -o r1 -- $B > /dev/null 2> $OUT/pmc_sq2.err
python $R/tools/rocpd_summary.py stats $(find $OUT/kt -name "*.db" | head -1) > $OUT/kernel_stats.csv
python $R/tools/rocpd_summary.py pmc $(find $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq1 $OUT/pmc_sq2 -name "*.db") > $OUT/pmc.csv
find $OUT -name "*.db" -delete
cat $OUT/run.log; head -4 $OUT/kernel_stats.csv; grep -E "k_render|k_first|k_continue" $OUT/pmc.csv
