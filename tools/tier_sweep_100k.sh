# development aid (GPU box): LDS tier sets at 100k reads (half-size key tables), ms/step and row kernels
cd ${GRAFT_REPO_ROOT:-/root/repo}
for T in "300,526,800,1064,1600,2105,2789,3500,4096,6144,8192,11008" "300,524,800,1064,1600,2100,2480,2860,3500,4096,6144,8192,11008" "300,524,800,1064,1600,2100,2860,4096,6144,8192,11008" "524,1064,2100,2860,4096,6144,8192,11008" "300,524,800,1064,1400,1750,2100,2480,2860,3200,3600,4096,11008"; do
BELLA_HIP_TIERS=$T python bench.py --reads 100000 --steps 6 --warmup 2 --no-cpu-baseline --no-xdrop 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$T', round(d['ms_per_step'],3), round(d['phases_ms_per_step']['row_kernels'],3), d['config']['pairs'])"
done
