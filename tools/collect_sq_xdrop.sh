#!/usr/bin/env bash
# SQ counters of the X-drop kernel (one pass, --kernel-trace only).  Run on the GPU box: bash tools/collect_sq_xdrop.sh
set -uo pipefail
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/sqx; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/p -o t -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-10k ${BENCH_ARGS:---reads 10000} --no-dropin --no-hifi --no-layout-ab > $OUT/bench.json 2> $OUT/bench.err || true
python - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(glob.glob("$OUT/p/*counter_collection.csv")[0])):
    k = r["Kernel_Name"].split("(")[0]
    if "xdrop" not in k and "kmers" not in k and "k_lookup" not in k and "k_emit" not in k and "k_write_tuples" not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
dur = collections.defaultdict(float)
for r in csv.DictReader(open(glob.glob("$OUT/p/*kernel_trace.csv")[0])):
    dur[r["Kernel_Name"].split("(")[0]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
with open("$OUT/summary.txt", "w") as o:
    for k, v in agg.items():
        line = "%s | %.1f us | " % (k, dur[k]) + " ".join("%s=%.4g" % (c, x) for c, x in sorted(v.items()))
        print(line); o.write(line + "\n")
PY
python -c "import json; d=json.loads([l for l in open(\"$OUT/bench.json\") if l.startswith(\"{\")][-1]); print(json.dumps(d.get(\"xdrop\")))"
