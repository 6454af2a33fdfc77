#!/bin/bash
# Build the C-ABI library of another revision next to the current one, for same-box A/B runs:
#   tools/build_variant.sh <git-rev> <name>   ->  multiview_motion_capture_amd/lib/libmvmc_<name>.so   (use with MVMC_LIB_PATH)
set -e
REV=$1; NAME=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d /tmp/mvmc_variant.XXXXXX)
git -C "$ROOT" archive "$REV" multiview_motion_capture_amd/csrc include | tar -x -C "$TMP"
mkdir -p "$TMP/multiview_motion_capture_amd/lib"
make -C "$TMP/multiview_motion_capture_amd/csrc" -j4 >/dev/null 2>&1
cp "$TMP/multiview_motion_capture_amd/lib/libmvmc_hip.so" "$ROOT/multiview_motion_capture_amd/lib/libmvmc_$NAME.so"
rm -rf "$TMP"
echo "built $ROOT/multiview_motion_capture_amd/lib/libmvmc_$NAME.so from $REV"
