#!/usr/bin/env bash
# Regenerates the judged artefacts of a round on the GPU box: the bench line, rocprofv3 kernel stats + the step timeline of the
# same commands (10k and 100k reads), HBM traffic (PMC) and SQ counters.  usage: bash tools/refresh_profiles.sh r05
# (outputs under gpurun_out/refresh/; copy into profiles/)
set -uo pipefail
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/refresh; rm -rf $OUT; mkdir -p $OUT
cd $R
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
for KEY in 10k 100k; do
  if [ $KEY = 10k ]; then ARGS="--reads 10000 --steps 20 --warmup 3"; else ARGS="--reads 100000 --steps 5 --warmup 2"; fi
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$KEY -o s -- python $R/bench.py $ARGS --no-cpu-baseline --no-10k --no-xdrop --no-dropin --no-hifi --no-layout-ab > $OUT/${TAG}_bench_under_rocprof_$KEY.json 2>/dev/null )
  python tools/summarize_rocprof.py $OUT/prof_$KEY/s_kernel_stats.csv 30 > $OUT/${TAG}_kernel_stats_$KEY.txt
  cp $OUT/prof_$KEY/s_kernel_stats.csv $OUT/${TAG}_kernel_stats_$KEY.csv
  python tools/timeline.py $OUT/prof_$KEY/s_kernel_trace.csv > $OUT/${TAG}_step_timeline_$KEY.txt
  rm -rf $OUT/prof_$KEY
done
bash tools/collect_traffic.sh 10k > $OUT/traffic_10k.log 2>&1
bash tools/collect_traffic.sh 100k --reads 100000 > $OUT/traffic_100k.log 2>&1
python - <<PY
import json
d = {k: json.load(open("$R/gpurun_out/traffic/%s.json" % k)) for k in ("10k", "100k")}
json.dump(d, open("$OUT/${TAG}_hbm_traffic.json", "w"), indent=1)
PY
bash tools/collect_sq.sh > $OUT/${TAG}_sq_counters_10k.txt 2>/dev/null
BENCH_ARGS="--reads 100000" bash tools/collect_sq.sh > $OUT/${TAG}_sq_counters_100k.txt 2>/dev/null
bash tools/collect_sq_xdrop.sh > $OUT/${TAG}_xdrop_sq.txt 2>/dev/null
# issue rates of the instruction kinds the kernels are priced against, LDS operation costs
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/valu_rates tools/ubench/valu_rates.hip && /tmp/valu_rates > $OUT/${TAG}_valu_rates.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/lds_ops tools/ubench/lds_ops.hip && /tmp/lds_ops > $OUT/${TAG}_lds_ops.txt 2>&1
# the HiFi-like probe (columns above the LDS tiers on the sort-based path) and one rank's share of the strong-scaling run
( python tools/hifi_probe2.py 3000 40 1 2>&1 | tail -1; bash tools/hifi_trace.sh 3000 40 1 2>&1 | tail -18 ) > $OUT/${TAG}_hifi_probe.txt
( python tools/hifi_probe2.py 10000 40 1 2>&1 | tail -1; bash tools/hifi_trace.sh 10000 40 1 2>&1 | tail -18 ) > $OUT/${TAG}_hifi_probe_10k.txt
python tools/rank_probe.py 2 4 8 2>&1 | tail -8 > $OUT/${TAG}_rank_probe_100k.txt
# the front end kernel by kernel, and the N > 1 branch of bench.py end to end on this one GPU (two ranks, gloo rendezvous)
bash tools/frontend_trace.sh 100000 > $OUT/${TAG}_frontend_100k.txt 2>&1
BELLA_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 2> $OUT/${TAG}_bench_gloo2_one_gpu.err | grep "^{" > $OUT/${TAG}_bench_gloo2_one_gpu.json
tail -n 2 $OUT/traffic_10k.log; tail -n 2 $OUT/traffic_100k.log; cat $OUT/${TAG}_bench.json | cut -c1-600; cat $OUT/${TAG}_step_timeline_10k.txt | tail -3
