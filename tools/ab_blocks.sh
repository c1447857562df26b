# development aid (GPU box): workgroup sizes per LDS class, variant libraries built with -DBELLA_CLASS_BLOCKS={...} under tools/_old/v/
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { for R in 100000 10000; do python bench.py --reads $R --steps $([ $R = 10000 ] && echo 30 || echo 6) --warmup 2 --no-cpu-baseline --no-xdrop 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', $R, round(d['ms_per_step'],4), round(d['phases_ms_per_step']['row_kernels'],4), d['config']['pairs'])"; done; }
run base
cp bella_amd/libbella_hip.so /tmp/new.so
for f in tools/_old/v/*.so; do cp $f bella_amd/libbella_hip.so; run $(basename $f); done
cp /tmp/new.so bella_amd/libbella_hip.so
run base_again
