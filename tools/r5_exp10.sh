#!/usr/bin/env bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5m; mkdir -p $O
cp bella_amd/libbella_hip.so /tmp/prod.so; cp tools/_old/wf_b640.so bella_amd/libbella_hip.so
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "medium_synthetic or half_size or ecsample or config1_full or out_of_order or key_table" 2>&1 | tail -3 | tee $O/tests.txt
cp /tmp/prod.so bella_amd/libbella_hip.so
TAG=r5m TENK=1 REPS=2 bash tools/r5_ab.sh
