#!/usr/bin/env bash
# development loop on the GPU box: core parity tests, then the 100k and 10k benches (short), optionally the per-phase profile.
# usage: bash tools/exp.sh [prof]
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "spgemm_pairs_bit_exact or medium_synthetic or big_lds or half_size or out_of_order or key_table_overflow or many_bins or config1_full or ecsample" 2>&1 | tail -6
python bench.py --reads 100000 --steps 5 --warmup 2 --no-cpu-baseline --no-xdrop --no-dropin --no-hifi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('100k ms/step %.3f rows %.3f frac %.4f pairs %d'%(d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'], d['config']['pairs']))"
python bench.py --reads 10000 --steps 20 --warmup 3 --no-cpu-baseline --no-xdrop --no-10k --no-dropin --no-hifi 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('10k ms/step %.4f rows %.4f frac %.4f pairs %d'%(d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'], d['config']['pairs']))"
if [ "${1:-}" = prof ]; then bash tools/prof_run.sh --reads 100000 --no-xdrop 2>&1 | tail -8; fi
