#!/usr/bin/env bash
# development aid: builds tools/_old/libbella_prof.so = the library with -DBELLA_DEV_PROF (per-phase cycle counters of the row
# kernels, printed to stderr after every pass).  Never shipped: the product library contains no such code.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_old
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-result -Wno-pass-failed -DBELLA_DEV_PROF \
    -o tools/_old/libbella_prof.so bella_amd/csrc/bella_hip.hip
