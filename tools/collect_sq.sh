#!/usr/bin/env bash
# SQ counters of the SpGEMM row kernels (one pass, --kernel-trace only): where the wave cycles go.
# Run on the GPU box:  bash tools/collect_sq.sh   (writes gpurun_out/sq/)
set -euo pipefail
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/sq
mkdir -p "$OUT"
CTRS="${CTRS:-SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS}"
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d "$OUT/p" -o t -- python "$R/bench.py" --steps 4 --warmup 1 --no-cpu-baseline --no-10k --no-xdrop --no-dropin --no-hifi --no-layout-ab ${BENCH_ARGS:---reads 10000} > "$OUT/bench.json" 2> "$OUT/bench.err" || true
python - <<PY
import csv, collections, glob
f = glob.glob("$OUT/p/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0]
    if not any(t in k for t in ("k_spgemm_rows", "k_row_flops", "k_tier_lists", "k_order_", "k_fold_overflow")): continue   # the pass
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k, r["Counter_Name"])] += 1
with open("$OUT/summary.txt", "w") as o:
    for k, v in agg.items():
        line = k + " | " + " ".join("%s=%.4g" % (c, x) for c, x in sorted(v.items())) + " | dispatches=%d" % max(n[(k, c)] for c in v)
        print(line); o.write(line + "\n")
PY
