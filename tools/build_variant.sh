#!/bin/bash
# Diagnostics: a build of the library with extra compile-time definitions, into build/libzultra_amd_<name>.so (A/B runs: tools/ab_lib.py).
# usage: tools/build_variant.sh <name> [-DX=Y ...]
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -x hip "$@" -I zultra_amd/csrc -o build/libzultra_amd_$name.so zultra_amd/csrc/zh_device.hip zultra_amd/csrc/libzultra.cpp
echo build/libzultra_amd_$name.so
