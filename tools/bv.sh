#!/usr/bin/env bash
# development aid: builds tools/_old/wf_NAME.so = the library with extra -D flags (variants for tools/r5_ab.sh).  usage: tools/bv.sh NAME [-DX ...]
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_old
N=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-result -Wno-pass-failed "$@" \
    -o tools/_old/wf_$N.so bella_amd/csrc/bella_hip.hip
