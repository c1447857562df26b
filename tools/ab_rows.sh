#!/usr/bin/env bash
# development aid (GPU box): same-box A/B of the row kernels at 100k reads: every tools/_old/wf_*.so against the current library, twice
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { python bench.py --reads ${READS:-100000} --steps 6 --warmup 2 --no-cpu-baseline --no-xdrop --no-dropin --no-hifi 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['phases_ms_per_step'].items()})"; }
cp bella_amd/libbella_hip.so /tmp/cur.so
for rep in 1 2; do
  for f in $(ls tools/_old/wf_*.so 2>/dev/null) /tmp/cur.so; do cp $f bella_amd/libbella_hip.so; run $(basename $f); done
done
cp /tmp/cur.so bella_amd/libbella_hip.so
