#!/bin/bash
# Code-object metadata of every kernel of bio_ik_amd/libbioik_hip.so (registers, spilled registers, scratch bytes per lane, LDS): extracts the gfx950 code
# object from the library's fat binary and reads its notes with llvm-readelf.  No GPU needed.  usage: tools/kernel_metadata.sh [library] > profiles/rNN_kernel_metadata.txt
LIB=${1:-bio_ik_amd/libbioik_hip.so}
T=$(mktemp -d)
L=/opt/rocm/lib/llvm/bin
$L/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin "$LIB" /dev/null 2>/dev/null || $L/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin "$LIB"
$L/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/k.co
echo "# $(basename $LIB): llvm-readelf --notes of the gfx950 code object ($(sha256sum "$LIB" | cut -c1-16)); sources $(cd $(dirname $0)/.. && python -c 'import bench; print(bench.kernel_sources_hash()[:16])' 2>/dev/null)"
printf "%-28s %6s %12s %6s %12s %8s\n" kernel vgprs vgpr_spills sgprs sgpr_spills scratch_B
$L/llvm-readelf --notes $T/k.co | awk '
/\.name:/ {name=$2}
/\.private_segment_fixed_size:/ {scr=$2}
/\.sgpr_count:/ {sg=$2}
/\.sgpr_spill_count:/ {sgs=$2}
/\.vgpr_count:/ {vg=$2}
/\.vgpr_spill_count:/ {vgs=$2; cmd="c++filt " name; cmd | getline dn; close(cmd); sub(/\(.*/,"",dn); printf "%-28s %6s %12s %6s %12s %8s\n", dn, vg, vgs, sg, sgs, scr}'
rm -rf $T
