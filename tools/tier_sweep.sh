for T in "768,1280,2048,3072,4096,6144,8192" "768,1280,2048,2688,3328,4096,8192" "768,1280,2048,3072,4096,6144,8192" "768,1280,2048,2688,3328,4096,8192"; do
BELLA_HIP_TIERS=$T python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$T', round(d['ms_per_step'],4), round(d['phases_ms_per_step']['row_kernels'],4))"
done
