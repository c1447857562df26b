#!/usr/bin/env bash
# round-5 experiment 1 (GPU box): parity of the layouts touched, the new default bench line end to end, first-appearance + inline A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "spgemm_pairs_bit_exact or partition_union or library_communicator or collectives or failing_rank or symbolic" 2>&1 | tail -4 > $O/tests.txt
run() { python bench.py --reads 100000 --steps 5 --warmup 2 --no-cpu-baseline --no-xdrop --no-dropin --no-hifi --no-layout-ab "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', 'ms/step %.3f rows %.3f order %.3f frac %.4f kcount %s asm %.1f (layout %.1f) reserve %s'%(d['ms_per_step'], d['phases_ms_per_step']['row_kernels'], d['phases_ms_per_step']['slot_order+placement'], d['roofline']['frac'], d['kcount']['runs_ms'], d['assemble']['ms'], d['assemble']['layout_ms'], d.get('reserve_ms')))"; }
run > $O/ab.txt 2>&1
run --debug-flags 1024 >> $O/ab.txt 2>&1
run >> $O/ab.txt 2>&1
run --debug-flags 1024 >> $O/ab.txt 2>&1
BELLA_BENCH_NO_RESERVE=1 run >> $O/ab.txt 2>&1
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt
cat $O/tests.txt $O/ab.txt; tail -3 $O/bench_time.txt; cut -c1-1500 $O/bench_default.json; tail -5 $O/bench_default.err
