#!/usr/bin/env bash
# HBM traffic of the SpGEMM row kernels from the PMC counters (MI355X_MICROARCH.md "HBM"): FETCH_SIZE and WRITE_SIZE in SEPARATE
# passes, each with --kernel-trace only.  Run on the GPU box:  bash tools/collect_traffic.sh <key> [bench args]
# (writes gpurun_out/traffic/<key>.json; key = "10k" | "100k")
set -euo pipefail
KEY=${1:-10k}; shift || true
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/traffic
mkdir -p "$OUT"; rm -rf "$OUT/$KEY"_*
STEPS=4
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/${KEY}_$C" -o t -- python "$R/bench.py" --steps $STEPS --warmup 1 --no-cpu-baseline --no-10k --no-xdrop --no-dropin --no-hifi --no-layout-ab ${@:---reads 10000} > "$OUT/${KEY}_bench_$C.json" 2> "$OUT/${KEY}_bench_$C.err" || true
done
python - <<PY
import csv, collections, json, glob
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$OUT/${KEY}_%s/*counter_collection.csv" % c)
    if not f: continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"].split("(")[0].split("<")[0]
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    out[c] = {k: {"sum_kb": v[0], "dispatches": v[1]} for k, v in agg.items() if "bella::" in k}
bench = json.loads([l for l in open("$OUT/${KEY}_bench_FETCH_SIZE.json") if l.startswith("{")][-1])
passes = $STEPS + 1
def tot(c, pred):
    return sum(v["sum_kb"] for k, v in out.get(c, {}).items() if pred(k)) * 1024 / passes
sp = lambda k: "k_spgemm_rows" in k or "k_fold" in k or "k_order" in k
fetch, write = tot("FETCH_SIZE", sp), tot("WRITE_SIZE", sp)
# calibration of FETCH_SIZE on our own coalesced stream: k_layout_live (assembly; the bench counts twice, assembles once: one dispatch
# in the run) reads the 8-byte B' entry of every nonzero once
nnz = bench["config"]["nnzA"]
rf = sum(v["sum_kb"] for k, v in out.get("FETCH_SIZE", {}).items() if "k_layout_live" in k) * 1024
summary = {
 "workload": "%d reads, 1 GPU" % bench["config"]["reads"], "layout": bench["roofline"].get("layout", "default"),
 "kernels": "k_spgemm_rows_lds (all LDS classes) + k_fold_overflow + k_order_wave/_block", "per": "step (= one launch set)",
 "FETCH_SIZE_raw_bytes": fetch, "WRITE_SIZE_raw_bytes": write,
 "fetch_calibration": {"kernel": "k_layout_live", "expected_bytes": 8 * nnz, "ratio_measured_over_expected": rf / (8.0 * nnz),
                       "note": "FETCH_SIZE reads 1/2 of the streamed bytes on gfx950 (MI355X_MICROARCH.md, HBM); checked on our own coalesced 8-byte stream"},
 "hbm_bytes_corrected": 2.0 * fetch + write,
 "algorithmic_bytes": bench["roofline"]["algorithmic_bytes_per_step"],
 "raw": out,
}
json.dump(summary, open("$OUT/$KEY.json", "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "raw"}))
PY
