#!/bin/bash
# A library variant that differs in mvmc_geom.hip's compile flags only (same-box A/B of the triangulation kernels):
#   tools/geom_variant.sh <name> "<extra flags>"   ->  multiview_motion_capture_amd/lib/libmvmc_<name>.so   (use with MVMC_LIB_PATH)
set -e
NAME=$1; EXTRA=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/multiview_motion_capture_amd/csrc"
make >/dev/null 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function $EXTRA -c mvmc_geom.hip -o /tmp/mvmc_geom_$NAME.o 2>/dev/null
OBJS=$(ls ../lib/obj/*.o | grep -v mvmc_geom.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libmvmc_$NAME.so $OBJS /tmp/mvmc_geom_$NAME.o
echo "built libmvmc_$NAME.so ($EXTRA)"
