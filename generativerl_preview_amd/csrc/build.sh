#!/bin/bash
# Builds libgrlhip.so in-tree for gfx950 (cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=../libgrlhip.so
$HIPCC -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -Wall -Wno-unused-function \
  -o "$OUT" grlhip.hip plan.cpp
echo "built $(realpath $OUT)"
