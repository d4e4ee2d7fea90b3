#!/bin/bash
# dev tool: link a copy of the library whose pwx.o is compiled with extra -D flags (A/B runs through CIDNET_LIB_PATH)
#   tools/build_pwx_variant.sh NAME -DPWX_MAX_MTW=3 ...
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/bin /tmp/pwxv
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Xclang -target-feature -Xclang -packed-fp32-ops -mllvm -amdgpu-sdwa-peephole=0 \
  -fno-slp-vectorize -I include -I hvi-cidnet_amd/csrc "$@" -c hvi-cidnet_amd/csrc/pwx.hip -o /tmp/pwxv/pwx_$name.o 2>&1 | grep -v "recognized feature" || true
objs=$(ls hvi-cidnet_amd/csrc/build/*.o | grep -v "/pwx.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libcidnet_pwx_$name.so $objs /tmp/pwxv/pwx_$name.o
echo built tools/bin/libcidnet_pwx_$name.so
