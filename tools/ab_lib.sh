#!/usr/bin/env bash
# development aid: same-box A/B of the current library against an older build (tools/_old/*.so), S100k by default
R=${READS:-100000}
run() { python bench.py --reads $R --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['phases_ms_per_step'].items()})"; }
run new
cp bella_amd/libbella_hip.so /tmp/new.so
for f in tools/_old/*.so; do cp $f bella_amd/libbella_hip.so; run $(basename $f); done
cp /tmp/new.so bella_amd/libbella_hip.so
run new_again
