#!/usr/bin/env bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${TAG:-r5n}; mkdir -p $O
V=${VARIANT:-tools/_old/wf_walk128.so}
cp bella_amd/libbella_hip.so /tmp/prod.so; cp $V bella_amd/libbella_hip.so
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "spgemm_pairs_bit_exact or medium_synthetic or half_size or ecsample or config1_full or out_of_order or key_table or many_bins or hifi_syncmer or big_lds" 2>&1 | tail -3 | tee $O/tests.txt
cp /tmp/prod.so bella_amd/libbella_hip.so
TENK=1 REPS=2 bash tools/r5_ab.sh
