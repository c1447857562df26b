#!/usr/bin/env bash
# development aid (GPU box): bench with the profiling build (tools/prof_build.sh), prints the per-class phase shares
cp bella_amd/libbella_hip.so /tmp/prod.so; cp tools/_old/libbella_prof.so bella_amd/libbella_hip.so
python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" 2>&1 | grep "bella_hip prof" | tail -8
cp /tmp/prod.so bella_amd/libbella_hip.so
