#!/usr/bin/env bash
# round-6 A/B on the GPU box over bench.py --tune settings (same library): usage: tools/r6_tune_ab.sh "" "compact_b=1" ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${TAG:-r6ab}; mkdir -p $O
run() { timeout 300 python bench.py --reads ${READS:-100000} --steps 6 --warmup 2 --no-cpu-baseline --no-xdrop --no-dropin --no-hifi --no-layout-ab --no-10k $1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$1]', 'ms/step %.3f sym %.3f rows %.3f order %.3f frac %.4f layout %.2f asm %.2f'%(d['ms_per_step'], d['phases_ms_per_step']['symbolic+tiering'], d['phases_ms_per_step']['row_kernels'], d['phases_ms_per_step']['slot_order+placement'], d['roofline']['frac'], d['assemble']['layout_ms'], d['assemble']['ms']))"; }
for rep in $(seq 1 ${REPS:-2}); do
  for t in "$@"; do
    args=""; for kv in $t; do args="$args --tune $kv"; done
    run "$args"
  done
done 2>&1 | tee $O/ab.txt
