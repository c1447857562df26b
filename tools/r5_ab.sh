#!/usr/bin/env bash
# round-5 A/B on the GPU box: every tools/_old/wf_*.so, ${REPS:-2} times each, 100k reads (and 10k with TENK=1)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${TAG:-r5ab}; mkdir -p $O
run() { timeout 300 python bench.py --reads ${2:-100000} --steps 6 --warmup 2 --no-cpu-baseline --no-xdrop --no-dropin --no-hifi --no-layout-ab 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 ${2:-100000}', 'ms/step %.3f rows %.3f order %.3f frac %.4f'%(d['ms_per_step'], d['phases_ms_per_step']['row_kernels'], d['phases_ms_per_step']['slot_order+placement'], d['roofline']['frac']))"; }
cp bella_amd/libbella_hip.so /tmp/cur.so
for rep in $(seq 1 ${REPS:-2}); do
  for f in tools/_old/wf_*.so; do cp $f bella_amd/libbella_hip.so; run $(basename $f); [ -n "${TENK:-}" ] && run $(basename $f) 10000; done
done 2>&1 | tee $O/ab.txt
cp /tmp/cur.so bella_amd/libbella_hip.so
