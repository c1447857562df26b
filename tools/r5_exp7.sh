#!/usr/bin/env bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "spgemm_pairs_bit_exact or medium_synthetic or big_lds or half_size or out_of_order or key_table_overflow or many_bins or config1_full or ecsample or wide_columns or hifi_syncmer or above_the_lds or partition_union or staged or symbolic" 2>&1 | tail -4 | tee $O/tests.txt
TAG=r5i TENK=1 REPS=2 bash tools/r5_ab.sh
