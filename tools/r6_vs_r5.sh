#!/usr/bin/env bash
# same-box A/B of the round-5 tree (tools/_old/r05tree: git archive of 51f5b9a + its library) against the current one, 100k reads
# (make the tree first, in the container: mkdir -p tools/_old/r05tree && git archive 51f5b9a | tar -x -C tools/_old/r05tree, build its library with hipcc into tools/_old/r05tree/bella_amd/libbella_hip.so)
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { ( cd $1 && GRAFT_REPO_ROOT=$1 python bench.py --reads 100000 --steps 6 --warmup 2 --no-cpu-baseline --no-xdrop --no-dropin --no-hifi --no-layout-ab $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', 'ms/step %.3f sym %.3f rows %.3f order %.3f frac %.4f'%(d['ms_per_step'], d['phases_ms_per_step']['symbolic+tiering'], d['phases_ms_per_step']['row_kernels'], d['phases_ms_per_step']['slot_order+placement'], d['roofline']['frac']))" ); }
for rep in 1 2 3; do run $R/tools/_old/r05tree round5 ""; run $R round6 "--no-e2e"; done
