#!/usr/bin/env bash
# tuning aid: ms/step of the bench workload for a few LDS tier sets
for T in "768,1280,2048,3072,4096" "768,1280,2048,2752,3712,4096" "1024,2048,2752,3712,4096" "1280,2752,3712,4096" "2752,3712,4096" "768,1280,1792,2304,2752,3712,4096" "640,1024,1536,2048,2752,3712,4096"; do
  BELLA_HIP_TIERS=$T python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$T', round(d['ms_per_step'],4), d['phases_ms_per_step']['row_kernels'])"
done
