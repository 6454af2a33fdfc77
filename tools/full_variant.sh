#!/bin/bash
# A library variant of the WORKING TREE built with extra compile flags (same-box A/B of an #ifdef experiment):
#   tools/full_variant.sh <name> "<extra flags>"   ->  multiview_motion_capture_amd/lib/libmvmc_<name>.so   (use with MVMC_LIB_PATH)
set -e
NAME=$1; EXTRA=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/multiview_motion_capture_amd/csrc"
O=/tmp/mvmc_variant_$NAME; rm -rf $O; mkdir -p $O
SRCS=$(sed -n 's/^SRCS  *:= *//p' Makefile)
pids=""
for f in $SRCS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function $EXTRA -c $f -o $O/${f%.hip}.o 2>/dev/null &
  pids="$pids $!"
  if [ $(jobs -r | wc -l) -ge 6 ]; then wait -n; fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libmvmc_$NAME.so $O/*.o
echo "built libmvmc_$NAME.so ($EXTRA)"
