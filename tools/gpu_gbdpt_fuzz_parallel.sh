# The G-BDPT fuzz (tools/gpu_gbdpt_fuzz.py) in N processes side by side -- its time is the oracle's, one core each; the GPU box has many.
#   gpurun --timeout 1500 -- 'FIRST=900000 PER=400 N=12 SPECULAR=1 [ENDPOINTS=1] bash tools/gpu_gbdpt_fuzz_parallel.sh'
cd $GRAFT_REPO_ROOT
FIRST=${FIRST:-900000}; PER=${PER:-300}; N=${N:-8}; LIMIT=${LIMIT:-1200}
mkdir -p gpurun_out/fuzz
pids=""
for i in $(seq 0 $((N - 1))); do
  if [ -n "$SPECULAR" ]; then export GBDPT_FUZZ_SPECULAR=1; fi
  if [ -n "$ENDPOINTS" ]; then export GBDPT_FUZZ_ENDPOINTS=1; fi
  timeout -s KILL $LIMIT python tools/gpu_gbdpt_fuzz.py $((FIRST + i * PER)) $PER > gpurun_out/fuzz/fuzz_$i.log 2>&1 &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
for i in $(seq 0 $((N - 1))); do echo "== process $i (seeds $((FIRST + i * PER)) ..)"; tail -2 gpurun_out/fuzz/fuzz_$i.log | cut -c1-400; done
grep -l MISMATCH gpurun_out/fuzz/*.log || echo "no mismatch in any process"
