#!/usr/bin/env bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r5full
( time timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 ) > gpurun_out/r5full/gpu_tests.txt 2>&1
cat gpurun_out/r5full/gpu_tests.txt
bash tools/refresh_profiles.sh r05 2>&1 | tail -12
