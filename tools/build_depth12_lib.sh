# lib/libgdpt_hip_d12.so: the product library with the G-BDPT records sized for maxDepth 12 (round 4's cap) instead of 20 -- the A/B of what the larger
# records cost the default depth.  Run AFTER the product build (links the other units' product objects); select with GDPT_LIB.
set -e
cd $(dirname $0)/..
P=gradientdomain-mitsuba_amd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wno-unused-value -Iinclude \
  -mllvm -amdgpu-spill-vgpr-to-agpr=0 -mllvm -amdgpu-function-calls=0 -DGDPT_BD_MAX_DEPTH=12 -c -o $P/lib/obj/gbdpt_capi_d12.o $P/csrc/gbdpt_capi.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/lib/libgdpt_hip_d12.so $P/lib/obj/gbdpt_capi_d12.o $(ls $P/lib/obj/*.o | grep -v "_O1.o\|_prof.o\|_d12.o\|gbdpt_capi.o\|gpt_wave_capi.o")
ls -la $P/lib/libgdpt_hip_d12.so
