#!/usr/bin/env bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5h; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "spgemm_pairs_bit_exact or medium_synthetic or half_size or many_bins or ecsample or golden_files or partition_union" 2>&1 | tail -4 | tee $O/tests.txt
TAG=r5h TENK=1 REPS=2 bash tools/r5_ab.sh
