#!/bin/bash
# Build libgrlhip_<name>.so with another engine_batch.hip.inc (same-box A/B timing: GRLHIP_LIB selects the library).
#   scripts/build_variant.sh /path/to/engine_batch.hip.inc name
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/generativerl_preview_amd/csrc
T=$(mktemp -d)
cp $C/*.hip $C/*.inc $C/*.hpp $C/*.cpp $T/
mkdir -p $T/../../include && cp $R/include/grlhip.h $T/../../include/ 2>/dev/null || true
cp "$1" $T/engine_batch.hip.inc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -mllvm -pragma-unroll-threshold=1048576 -I$R/include -c $T/grlhip.hip -o $T/grlhip.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/generativerl_preview_amd/libgrlhip_$2.so $T/grlhip.o $(ls $C/build/*.o | grep -v grlhip.o)
rm -rf $T
echo built libgrlhip_$2.so
