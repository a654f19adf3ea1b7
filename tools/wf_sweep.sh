python tools/gpu_wf_check.py check 1,6 > gpurun_out/wf2.log 2>&1; tail -1 gpurun_out/wf2.log
export WF_PIPE=3 GDPT_WF_ITERS=1
for cfg in "64 64" "64 32" "64 16" "64 8" "32 16" "16 16" "16 8" "32 8" "48 24"; do set -- $cfg
  export GDPT_WF_REFILL=$1 GDPT_WF_LEAFMIN=$2
  timeout 300 bash tools/kt_list.sh python tools/gpu_one_render.py atrium 8 > gpurun_out/wf_kt.log 2>&1
  echo "refill $1 leafMin $2: $(grep k_wf_trace gpurun_out/wf_kt.log | tail -2 | awk '{print $(NF-1)}' | tr '\n' ' ')  $(grep 'atrium 8 spp' gpurun_out/wf_kt.log | tail -1)"
done
