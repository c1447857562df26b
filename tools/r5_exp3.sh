#!/usr/bin/env bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5e; mkdir -p $O
( free -g; nproc; df -h /tmp | tail -1 ) > $O/box.txt 2>&1; cat $O/box.txt
cp bella_amd/libbella_hip.so /tmp/prod.so; cp tools/_old/libbella_prof.so bella_amd/libbella_hip.so
run() { python bench.py --reads 100000 --steps 4 --warmup 1 --no-cpu-baseline --no-xdrop --no-dropin --no-hifi --no-layout-ab 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skip=$1', 'ms/step %.3f rows %.3f pairs %d'%(d['ms_per_step'], d['phases_ms_per_step']['row_kernels'], d['config']['pairs']))"; }
for m in 0 1 3 7 24 16 8; do BELLA_DEV_SKIP_CLASSES=$m run $m; done 2>&1 | tee $O/skip.txt
cp /tmp/prod.so bella_amd/libbella_hip.so
timeout 600 python tools/dev/hifi_big_probe.py 300000 8 2>&1 | tee $O/hifi300k.txt
