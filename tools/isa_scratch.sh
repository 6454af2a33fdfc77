#!/bin/bash
# Register / scratch summary of the chain kernel's translation unit (no GPU needed):
#   bash tools/isa_scratch.sh [mvmc_chain.hip]
# Prints the kernel descriptors (VGPRs, scratch bytes per lane) and, per function, the static count of scratch stores / loads and
# of global stores -- the sources of the HBM-side traffic DESIGN.md section 6a accounts for.
set -e
SRC=${1:-mvmc_chain.hip}
cd "$(dirname "$0")/../multiview_motion_capture_amd/csrc"
OUT=/tmp/isa_$(basename $SRC .hip).s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $EXTRA -S --cuda-device-only -o $OUT $SRC 2>/dev/null
echo "ISA in $OUT"
awk '/^[ \t]*\.amdhsa_kernel /{k=$2} /amdhsa_next_free_vgpr|amdhsa_private_segment_fixed_size|amdhsa_group_segment_fixed_size/{print k, $1, $2}' $OUT | grep chain_kernel
awk '/^[_A-Za-z0-9$.]+:/{ if ($1 !~ /^\.L/) {fn=$1} }
     /scratch_store/{st[fn]++; seen[fn]=1} /scratch_load/{ld[fn]++} /global_store/{gs[fn]++} /s_swappc/{cl[fn]++}
     END{for (f in seen) printf "%6d st %6d ld %5d gst %3d calls  %s\n", st[f], ld[f], gs[f], cl[f], substr(f,1,90)}' $OUT | sort -rn | head -25
