# Code-object metadata (registers, spills, scratch bytes per lane) of the kernels of one translation unit's object:
#   bash tools/kernel_meta.sh gradientdomain-mitsuba_amd/lib/obj/gbdpt_capi.o [name-filter]
set -e
O=$(readlink -f "$1"); T=$(mktemp -d); cd $T
L=/opt/rocm/lib/llvm/bin
$L/llvm-objcopy --dump-section .hip_fatbin=fat.bin "$O" /dev/null
$L/clang-offload-bundler --unbundle --type=o --input=fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=co
$L/llvm-readelf --notes co | grep -E "^\s+\.name:|\.vgpr_count|\.agpr_count|private_segment_fixed_size|\.vgpr_spill_count|\.sgpr_count" | sed 's/^ *//;s/^- //' | paste - - - - - - \
  | awk '{for(i=1;i<=NF;i++) if($i==".name:") n=$(i+1); gsub(/\.name: *[^\t]*\t?/,""); print n "\t" $0}' | c++filt | sed 's/(.*)\t/\t/' | grep -E "${2:-.}" | sed 's/  */ /g'
rm -rf $T
