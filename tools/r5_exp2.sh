#!/usr/bin/env bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "spgemm_pairs_bit_exact or medium_synthetic or big_lds or half_size or out_of_order or key_table_overflow or many_bins or config1_full or ecsample or wide_columns or hifi_syncmer or above_the_lds" 2>&1 | tail -6 > $O/tests.txt
cat $O/tests.txt
run() { python bench.py --reads 100000 --steps 6 --warmup 2 --no-cpu-baseline --no-xdrop --no-dropin --no-hifi --no-layout-ab "${@:2}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms/step %.3f rows %.3f order %.3f frac %.4f asm %.1f (layout %.1f)'%(d['ms_per_step'], d['phases_ms_per_step']['row_kernels'], d['phases_ms_per_step']['slot_order+placement'], d['roofline']['frac'], d['assemble']['ms'], d['assemble']['layout_ms']))"; }
cp bella_amd/libbella_hip.so /tmp/cur.so
for rep in 1 2; do
  for f in tools/_old/wf_base.so /tmp/cur.so; do cp $f bella_amd/libbella_hip.so; run $(basename $f); done
done 2>&1 | tee $O/ab.txt
cp /tmp/cur.so bella_amd/libbella_hip.so
run cur_firstapp --layout-debug 1024 2>&1 | tee -a $O/ab.txt
run cur_firstapp --layout-debug 1024 2>&1 | tee -a $O/ab.txt
python bench.py --reads 10000 --steps 20 --warmup 3 --no-cpu-baseline --no-xdrop --no-dropin --no-hifi --no-layout-ab 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('10k cur ms/step %.4f rows %.4f'%(d['ms_per_step'], d['phases_ms_per_step']['row_kernels']))" | tee -a $O/ab.txt
cp tools/_old/wf_base.so bella_amd/libbella_hip.so
python bench.py --reads 10000 --steps 20 --warmup 3 --no-cpu-baseline --no-xdrop --no-dropin --no-hifi --no-layout-ab 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('10k base ms/step %.4f rows %.4f'%(d['ms_per_step'], d['phases_ms_per_step']['row_kernels']))" | tee -a $O/ab.txt
cp /tmp/cur.so bella_amd/libbella_hip.so
