#!/usr/bin/env bash
# Regenerates the judged artefacts of a round on the GPU box: bench line, rocprofv3 kernel stats of the same command, HBM traffic.
# usage: bash tools/refresh_profiles.sh r01_c      (outputs under gpurun_out/refresh/, copy into profiles/)
set -uo pipefail
TAG=${1:-r01_x}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/refresh; rm -rf $OUT; mkdir -p $OUT
cd $R
python bench.py --steps 20 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o s -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_bench_under_rocprof.json 2>/dev/null )
python tools/summarize_rocprof.py $OUT/prof/s_kernel_stats.csv 30 > $OUT/${TAG}_kernel_stats.txt
cp $OUT/prof/s_kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
bash tools/collect_traffic.sh > $OUT/traffic.log 2>&1
cp $R/gpurun_out/traffic/hbm_traffic.json $OUT/r01_hbm_traffic.json
cp $R/gpurun_out/traffic/traffic_raw.json $OUT/r01_hbm_traffic_raw.json
rm -rf $OUT/prof
tail -3 $OUT/traffic.log; cat $OUT/${TAG}_bench.json; head -12 $OUT/${TAG}_kernel_stats.txt
