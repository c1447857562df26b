#!/usr/bin/env bash
# development aid (GPU box): memory-pipeline counters of the SpGEMM row kernels (vector cache -> L2 -> fabric), two --pmc passes.
# usage: BENCH_ARGS="--reads 100000" bash tools/collect_mem.sh > gpurun_out/mem.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for SET in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_avr" "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_TAG_STALL_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  OUT=$R/gpurun_out/memc; rm -rf $OUT; mkdir -p $OUT
  timeout ${PER_RUN_TIMEOUT:-240} rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-xdrop --no-10k --no-dropin --no-hifi --no-layout-ab ${BENCH_ARGS:---reads 10000} > /dev/null 2>&1
  python - <<PY
import csv, collections, glob, sys
if not glob.glob("$OUT/*counter_collection.csv"): print("(no counters: timed out) $SET"); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(glob.glob("$OUT/*counter_collection.csv")[0])):
    k = r["Kernel_Name"]
    if "k_spgemm_rows_lds" not in k and "k_order_wave" not in k: continue
    k = k[k.index("k_"):k.index(">") + 1] if ">" in k else k
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, v in sorted(agg.items()):
    print(k, "dispatches", max(n[(k, c)] for c in v), " ".join("%s=%.4g" % (c, x) for c, x in sorted(v.items())))
PY
done
