#!/bin/bash
# timing experiments: xbuild.sh <name> [ENV=VAL ...]  -> ../xlib_<name>.so (git- and gpurun-ignored unless named in the run; delete after the experiment) with the bench kernel (b16_0_0_0) rebuilt under the given generator environment
set -e
cd "$(dirname "$0")"
name=$1; shift
rm -rf build_$name; cp -r build build_$name
env GEN_EXPERIMENT=1 "$@" python3 gen_bf16_body.py build_$name/engine_bf16_body_gen.inc 2> build_$name/gen.log
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -mllvm -pragma-unroll-threshold=1048576 -DB16_BODY_FILE="\"build_$name/engine_bf16_body_gen.inc\"" -DB16_EXPERIMENT -DB16_VARIANT=0 -DB16_EXTRA=0 -DB16_GUIDE=0 $XDEFS -c engine_bf16_inst.hip -o build_$name/b16_0_0_0.o 2>&1 | grep -E "error" || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../xlib_$name.so build_$name/*.o
tail -2 build_$name/gen.log | head -1
