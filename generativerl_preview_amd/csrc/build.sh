#!/bin/bash
# Builds libgrlhip.so in-tree for gfx950 (cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
make -j6 "$@"
echo "built $(realpath ../libgrlhip.so)"
