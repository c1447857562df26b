python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phases_ms_per_step'], d['roofline']['frac'])"
BELLA_HIP_PHASE_TIMERS=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep "row-kernel" | tail -1
if [ -n "$BIG" ]; then BELLA_HIP_PHASE_TIMERS=1 python bench.py --reads $BIG --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | grep "row-kernel\|^{" | tail -2 | cut -c1-400; fi
