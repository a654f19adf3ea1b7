# The G-PT side's fuzz tools side by side on the final binary (their time is the oracle's, one core each):
#   gpurun --timeout 1500 -- 'FIRST=1300000 PER=500 bash tools/gpu_fuzz_all_parallel.sh'
# 4 x gpu_fuzz_campaign.py through the LDS-scene builds, 4 x through the HBM-scene builds (GDPT_SCENE_IN_HBM=1), 2 x gpu_fuzz_features.py, 2 x gpu_serial_fuzz.py
cd $GRAFT_REPO_ROOT
FIRST=${FIRST:-1300000}; PER=${PER:-500}; LIMIT=${LIMIT:-1300}
mkdir -p gpurun_out/fuzz_all
pids=""
for i in 0 1 2 3; do timeout -s KILL $LIMIT python tools/gpu_fuzz_campaign.py $((FIRST + i * PER)) $PER > gpurun_out/fuzz_all/lds_$i.log 2>&1 & pids="$pids $!"; done
for i in 0 1 2 3; do GDPT_SCENE_IN_HBM=1 timeout -s KILL $LIMIT python tools/gpu_fuzz_campaign.py $((FIRST + 10000 + i * PER)) $PER > gpurun_out/fuzz_all/hbm_$i.log 2>&1 & pids="$pids $!"; done
for i in 0 1; do timeout -s KILL $LIMIT python tools/gpu_fuzz_features.py $((FIRST + 20000 + i * PER)) $PER > gpurun_out/fuzz_all/features_$i.log 2>&1 & pids="$pids $!"; done
for i in 0 1; do timeout -s KILL $LIMIT python tools/gpu_serial_fuzz.py $((FIRST + 30000 + i * PER)) $((PER / 2)) > gpurun_out/fuzz_all/serial_$i.log 2>&1 & pids="$pids $!"; done
for p in $pids; do wait $p; done
for f in gpurun_out/fuzz_all/*.log; do echo "== $f"; tail -2 $f | cut -c1-500; done
grep -h "^FUZZ_SUMMARY" gpurun_out/fuzz_all/*.log | sed "s/^FUZZ_SUMMARY //" > gpurun_out/fuzz_all/summaries.jsonl; wc -l gpurun_out/fuzz_all/summaries.jsonl
