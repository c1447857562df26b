#!/usr/bin/env bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5f; mkdir -p $O
cp bella_amd/libbella_hip.so /tmp/prod.so
cp tools/_old/wf_b64_64.so bella_amd/libbella_hip.so
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "spgemm_pairs_bit_exact or medium_synthetic or half_size or out_of_order or key_table_overflow or many_bins or ecsample" 2>&1 | tail -4 | tee $O/tests64.txt
cp /tmp/prod.so bella_amd/libbella_hip.so
TAG=r5f TENK=1 REPS=2 bash tools/r5_ab.sh
cp /tmp/prod.so bella_amd/libbella_hip.so
( time timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -s -m gpu -k "hifi_1M or fewer_nonzeros or partition_union" 2>&1 | tail -12 ) 2>&1 | tee $O/hifi1m.txt
