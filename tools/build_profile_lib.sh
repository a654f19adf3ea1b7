# lib/libgdpt_hip_prof.so: the product library with gbdpt_capi.hip compiled -DGDPT_BD_PROFILE (lane clocks per section of k_bdg_offset, read by
# tools/gpu_gbdpt_profile.py through GDPT_LIB).  Run AFTER the product build (it links the other units' product objects).
set -e
cd $(dirname $0)/..
P=gradientdomain-mitsuba_amd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wno-unused-value -Iinclude \
  -mllvm -amdgpu-spill-vgpr-to-agpr=0 -mllvm -amdgpu-function-calls=0 -DGDPT_BD_PROFILE -c -o $P/lib/obj/gbdpt_capi_prof.o $P/csrc/gbdpt_capi.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/lib/libgdpt_hip_prof.so $P/lib/obj/gbdpt_capi_prof.o $(ls $P/lib/obj/*.o | grep -v "_O1.o\|_prof.o\|gbdpt_capi.o\|gpt_wave_capi.o")
ls -la $P/lib/libgdpt_hip_prof.so
