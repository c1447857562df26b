#!/usr/bin/env bash
# development aid (GPU box): the row kernels cut after each phase (profiling build, BELLA_DEV_STOP), plain timing -- no counters
cd ${GRAFT_REPO_ROOT:-/root/repo}
cp bella_amd/libbella_hip.so /tmp/cur.so; cp tools/_old/libbella_prof.so bella_amd/libbella_hip.so
for rep in 1 2; do
for S in 0 1 2 3 4 5 6 -1; do
  BELLA_DEV_STOP=$S python bench.py --reads ${READS:-100000} --steps 6 --warmup 2 --no-cpu-baseline --no-xdrop --no-dropin --no-hifi --no-layout-ab 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stop $S', 'ms/step %.3f rows %.3f order %.3f'%(d['ms_per_step'], d['phases_ms_per_step']['row_kernels'], d['phases_ms_per_step']['slot_order+placement']))"
done; done
cp /tmp/cur.so bella_amd/libbella_hip.so
