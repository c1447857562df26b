# development aid: 10k and 100k bench lines (ms/step, row kernels)
for R in 10000 100000; do python bench.py --reads $R --steps $([ $R = 10000 ] && echo 40 || echo 6) --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print($R, round(d['ms_per_step'],4), round(d['phases_ms_per_step']['row_kernels'],4))"; done
